// am_peaks.hip -- score scan + peak pick kernels of libaudiomatch_amd.so (gfx950).
//
// Replaces find_peaks() of the reference (audio_matcher.rs:221-230), i.e.
// find_peaks::PeakFinder::new(scores).with_min_prominence(p)
//     .with_min_distance(d).find_peaks(), run per reference chunk on the slice
// [a, b) of the haystack's global score array (audio_matcher.rs:119-126).
//
// Semantics (crate find_peaks 0.1, scipy-style; see oracle/oracle.c for the
// pinning status of each rule):
//   local maximum with flat top: x[i-1] < x[i] == ... == x[k-1] > x[k],
//   never at the first/last sample of the chunk; prominence = height -
//   max(left_min, right_min), each min taken outwards until a strictly higher
//   sample or the chunk edge; keep prominence >= min_prominence; then greedy
//   min_distance filter by descending height; output by descending height.
//
// Design: a min/max summary per 1024-score tile lets one workgroup per chunk
//   (1) get the chunk minimum, (2) skip every tile that cannot hold a
//   qualifying peak (prominence <= height - chunk_min), (3) walk the
//   prominence ranges tile-wise; raw scores are only touched in candidate
//   tiles and at the two ends of each walk.  Walks are wavefront-cooperative:
//   64 lanes look at 64 samples / 64 tile summaries per step and agree on the
//   stopping point with a ballot.
#include "am_kernels.h"

#include <float.h>

namespace am {

constexpr int kPeakThreads = 256;
constexpr int kWaves = kPeakThreads / 64;
constexpr int kQueueCap = kTile / 2;  // a piece of at most kTile scores has at most kTile / 2 flat-topped maxima
constexpr int kGroup = 8;         // tiles per lane in the coarse step of a prominence walk
constexpr int kCandCap = kWideTileList;    // candidate tiles listed per chunk before falling back to all tiles

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------
// `bad` (optional, host-visible): set to 1 when a score is not finite.  A NaN or an infinity among
// the samples of an overlap-save block poisons every score of the block pair it belongs to (the
// transforms spread it); fminf / fmaxf drop NaNs, so the summaries would hide them.
__device__ __forceinline__ bool not_finite(float v) { return !(fabsf(v) <= FLT_MAX); }
__global__ void __launch_bounds__(256) tile_stats(const float* __restrict__ g, long long n, float2* __restrict__ stats, int* bad) {
    __shared__ float smin[4], smax[4];
    const long long base = (long long)blockIdx.x * kTile;
    float mn = FLT_MAX, mx = -FLT_MAX;
    bool nf = false;
    for (int i = threadIdx.x; i < kTile; i += 256) {
        const long long idx = base + i;
        if (idx < n) { const float v = g[idx]; mn = fminf(mn, v); mx = fmaxf(mx, v); nf |= not_finite(v); }
    }
    if (nf && bad != nullptr) *bad = 1;
    mn = wave_min(mn); mx = wave_max(mx);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { smin[wv] = mn; smax[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { mn = fminf(mn, smin[k]); mx = fmaxf(mx, smax[k]); }
        stats[blockIdx.x] = make_float2(fminf(mn, smin[0]), fmaxf(mx, smax[0]));
    }
}

// level-1 summary (1024 scores) from K3's level-0 summary (32 scores): eight lanes
// per tile, each with four consecutive entries (two 16-byte loads)
// (a run of 32 scores that were all NaN has no ordered (min,max) pair; an infinite one shows in it)
__device__ __forceinline__ bool bad_run(float mn, float mx) { return !(mn <= mx) || not_finite(mn) || not_finite(mx); }
__global__ void __launch_bounds__(256) stats_reduce(const float2* __restrict__ s32, long long n32,
                                                    float2* __restrict__ stats, long long ntiles, int* bad, PickGroup grp) {
    if (grp.n > 0) { s32 = grp.stats32[blockIdx.y]; stats = grp.stats[blockIdx.y]; }   // (one launch for the needles of a group)
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long tile = gid >> 3;
    const int l = threadIdx.x & 7;
    float mn = FLT_MAX, mx = -FLT_MAX;
    const long long idx = tile * 32 + 4 * l;
    if (tile < ntiles) {
        if (idx + 3 < n32) {
            const float4 a = *reinterpret_cast<const float4*>(s32 + idx);
            const float4 b = *reinterpret_cast<const float4*>(s32 + idx + 2);
            mn = fminf(fminf(a.x, a.z), fminf(b.x, b.z));
            mx = fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w));
            if (bad != nullptr && (bad_run(a.x, a.y) || bad_run(a.z, a.w) || bad_run(b.x, b.y) || bad_run(b.z, b.w))) *bad = 1;
        } else {
            for (int k = 0; k < 4; ++k)
                if (idx + k < n32) {
                    const float2 v = s32[idx + k]; mn = fminf(mn, v.x); mx = fmaxf(mx, v.y);
                    if (bad != nullptr && bad_run(v.x, v.y)) *bad = 1;
                }
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (l == 0 && tile < ntiles) stats[tile] = make_float2(mn, mx);
}

// ---------------------------------------------------------------------------
struct Cand { long long ps, pe; float h; };

// K3 writes raw scores only for the 32-score runs that can matter (am_fft.hip, k3_finish);
// everywhere else only the per-32 summary exists.  The flags are the ballots of K3's wavefronts.
__device__ __forceinline__ bool run_written(const SparseScores& sp, long long idx) {
    if (sp.wbits == nullptr) return true;
    // idx / hop through one f64 multiply and a fix-up (idx < 2^50)
    long long blk = (long long)((double)idx * sp.inv_hop);
    long long rem = idx - blk * sp.hop;
    if (rem < 0) { rem += sp.hop; --blk; }
    else if (rem >= sp.hop) { rem -= sp.hop; ++blk; }
    const unsigned n = (unsigned)rem;
    const unsigned row = n >> sp.log_n2, tile = (n & ((1u << sp.log_n2) - 1u)) >> kColsLog;
    const int hb = sp.log_n1 - 4;                              // rows per register index a of a column owner
    const unsigned wave = (row >> 2) & ((1u << (hb - 2)) - 1u), bit = ((row >> hb) << 2) | (row & 3u);
    const long long word = (((blk << (sp.log_n2 - kColsLog)) + tile) << (hb - 2)) + wave;
    return (sp.wbits[word] >> bit) & 1ull;
}
// for minima: exact where written, else the run's minimum (exact whenever the
// whole run lies in the range being reduced, a lower bound otherwise)
__device__ __forceinline__ float score_for_min(const float* __restrict__ g, const SparseScores& sp, long long idx) {
    return run_written(sp, idx) ? g[idx] : sp.stats32[idx >> 5].x;
}
// for comparisons with a candidate height: an unwritten score lies below the write threshold of its
// tile, which the chunk's certificate (peaks_kernel) has shown to be below every candidate height
__device__ __forceinline__ float score_for_cmp(const float* __restrict__ g, const SparseScores& sp, long long idx) {
    return run_written(sp, idx) ? g[idx] : -FLT_MAX;
}

// One wave-cooperative step of the walk to the left of `cur` (exclusive) down
// to `a`: either skips up to 64 whole tiles through their summaries or looks at
// up to 64 raw samples.  Returns true when a strictly higher sample ended the
// walk; `cur` reaching `a` ends it at the chunk edge.
__device__ __forceinline__ bool step_left(const float* __restrict__ g, const float2* __restrict__ stats,
                                          const SparseScores& sp,
                                          long long a, long long& cur, float h, float& vmin, int lane) {
    // coarse skip: every lane summarises a group of kGroup tiles (64 groups per step)
    if ((cur % kTile) == 0 && cur - (long long)kGroup * kTile >= a) {
        const long long t1 = cur / kTile - (long long)kGroup * lane;      // group = tiles [t1 - kGroup, t1)
        const bool valid = (t1 - kGroup) * (long long)kTile >= a;
        float gmn = FLT_MAX, gmx = -FLT_MAX;
        if (valid) {
#pragma unroll
            for (int k = 1; k <= kGroup; ++k) { const float2 st = stats[t1 - k]; gmn = fminf(gmn, st.x); gmx = fmaxf(gmx, st.y); }
        }
        const unsigned long long blocked = __ballot(valid && gmx > h);
        const int nvalid = __popcll(__ballot(valid));
        const int nskip = blocked ? (__ffsll((long long)blocked) - 1) : nvalid;
        if (nskip > 0) {
            vmin = fminf(vmin, wave_min(lane < nskip ? gmn : FLT_MAX));
            cur -= (long long)nskip * kGroup * kTile;
            return false;
        }
    }
    if ((cur % kTile) == 0 && cur - kTile >= a) {
        const long long t = cur / kTile - 1 - lane;
        const bool valid = t >= 0 && t * (long long)kTile >= a;
        float2 st = make_float2(FLT_MAX, -FLT_MAX);
        if (valid) st = stats[t];
        const unsigned long long blocked = __ballot(valid && st.y > h);
        const int nvalid = __popcll(__ballot(valid));
        const int nskip = blocked ? (__ffsll((long long)blocked) - 1) : nvalid;
        if (nskip > 0) {
            vmin = fminf(vmin, wave_min(lane < nskip ? st.x : FLT_MAX));
            cur -= (long long)nskip * kTile;
            return false;
        }
    }
    const long long tile_lo = ((cur - 1) / kTile) * kTile;
    const long long lo = tile_lo > a ? tile_lo : a;
    // run-level skip through K3's exact (min,max) per 32 scores (stays inside the
    // current tile so that tile-level skipping resumes at its boundary)
    if (sp.stats32 != nullptr && (cur & 31) == 0 && cur - 32 >= lo) {
        const long long r = (cur >> 5) - 1 - lane;
        const bool valid = r * 32 >= lo;
        float2 st = make_float2(FLT_MAX, -FLT_MAX);
        if (valid) st = sp.stats32[r];
        const unsigned long long blocked = __ballot(valid && st.y > h);
        const int nvalid = __popcll(__ballot(valid));
        const int nskip = blocked ? (__ffsll((long long)blocked) - 1) : nvalid;
        if (nskip > 0) {
            vmin = fminf(vmin, wave_min(lane < nskip ? st.x : FLT_MAX));
            cur -= (long long)nskip * 32;
            return false;
        }
    }
    // A raw step stops at the start of the current 32-score run when run summaries
    // exist: the walk is then aligned for the run-level skip above (raw steps of 64
    // would keep the misalignment they started with all the way to the tile edge).
    const long long run_lo = (cur - 1) & ~31ll;
    const long long lo2 = (sp.stats32 != nullptr && run_lo > lo) ? run_lo : lo;
    const long long idx = cur - 1 - lane;
    const bool valid = idx >= lo2;
    const float v = valid ? score_for_min(g, sp, idx) : 0.0f;
    const unsigned long long higher = __ballot(valid && v > h);
    const int nval = __popcll(__ballot(valid));
    const int ntake = higher ? (__ffsll((long long)higher) - 1) : nval;
    vmin = fminf(vmin, wave_min(lane < ntake ? v : FLT_MAX));
    cur -= ntake;
    return higher != 0ull;
}

// Mirror image: walk to the right from `cur` (inclusive) up to `b` (exclusive).
__device__ __forceinline__ bool step_right(const float* __restrict__ g, const float2* __restrict__ stats,
                                           const SparseScores& sp,
                                           long long b, long long& cur, float h, float& vmin, int lane) {
    if ((cur % kTile) == 0 && cur + (long long)kGroup * kTile <= b) {
        const long long t0 = cur / kTile + (long long)kGroup * lane;      // group = tiles [t0, t0 + kGroup)
        const bool valid = (t0 + kGroup) * (long long)kTile <= b;
        float gmn = FLT_MAX, gmx = -FLT_MAX;
        if (valid) {
#pragma unroll
            for (int k = 0; k < kGroup; ++k) { const float2 st = stats[t0 + k]; gmn = fminf(gmn, st.x); gmx = fmaxf(gmx, st.y); }
        }
        const unsigned long long blocked = __ballot(valid && gmx > h);
        const int nvalid = __popcll(__ballot(valid));
        const int nskip = blocked ? (__ffsll((long long)blocked) - 1) : nvalid;
        if (nskip > 0) {
            vmin = fminf(vmin, wave_min(lane < nskip ? gmn : FLT_MAX));
            cur += (long long)nskip * kGroup * kTile;
            return false;
        }
    }
    if ((cur % kTile) == 0 && cur + kTile <= b) {
        const long long t = cur / kTile + lane;
        const bool valid = (t + 1) * (long long)kTile <= b;
        float2 st = make_float2(FLT_MAX, -FLT_MAX);
        if (valid) st = stats[t];
        const unsigned long long blocked = __ballot(valid && st.y > h);
        const int nvalid = __popcll(__ballot(valid));
        const int nskip = blocked ? (__ffsll((long long)blocked) - 1) : nvalid;
        if (nskip > 0) {
            vmin = fminf(vmin, wave_min(lane < nskip ? st.x : FLT_MAX));
            cur += (long long)nskip * kTile;
            return false;
        }
    }
    const long long tile_hi = (cur / kTile + 1) * kTile;
    const long long hi = tile_hi < b ? tile_hi : b;
    if (sp.stats32 != nullptr && (cur & 31) == 0 && cur + 32 <= hi) {
        const long long r = (cur >> 5) + lane;
        const bool valid = (r + 1) * 32 <= hi;
        float2 st = make_float2(FLT_MAX, -FLT_MAX);
        if (valid) st = sp.stats32[r];
        const unsigned long long blocked = __ballot(valid && st.y > h);
        const int nvalid = __popcll(__ballot(valid));
        const int nskip = blocked ? (__ffsll((long long)blocked) - 1) : nvalid;
        if (nskip > 0) {
            vmin = fminf(vmin, wave_min(lane < nskip ? st.x : FLT_MAX));
            cur += (long long)nskip * 32;
            return false;
        }
    }
    // as in step_left: end a raw step at the next run boundary when run summaries exist
    const long long run_hi = ((cur >> 5) + 1) << 5;
    const long long hi2 = (sp.stats32 != nullptr && run_hi < hi) ? run_hi : hi;
    const long long idx = cur + lane;
    const bool valid = idx < hi2;
    const float v = valid ? score_for_min(g, sp, idx) : 0.0f;
    const unsigned long long higher = __ballot(valid && v > h);
    const int nval = __popcll(__ballot(valid));
    const int ntake = higher ? (__ffsll((long long)higher) - 1) : nval;
    vmin = fminf(vmin, wave_min(lane < ntake ? v : FLT_MAX));
    cur += ntake;
    return higher != 0ull;
}

// Prominence of the flat-topped maximum [ps, pe) of height h inside chunk
// [a, b); both walks advance in lock step so that a side lobe next to a taller
// peak is rejected after a few samples (prominence <= h - min of a finished
// side).  Returns false when prominence < min_prom.
__device__ bool prominence(const float* __restrict__ g, const float2* __restrict__ stats,
                           const SparseScores& sp, long long a, long long b, long long ps, long long pe, float h,
                           float min_prom, int lane, float& prom) {
    long long cl = ps, cr = pe;
    float lmin = h, rmin = h;
    bool dl = cl <= a, dr = cr >= b;
    while (!dl || !dr) {
        if (!dl) {
            const bool stopped = step_left(g, stats, sp, a, cl, h, lmin, lane);
            dl = stopped || cl <= a;
            if (dl && !((h - lmin) >= min_prom)) return false;
        }
        if (!dr) {
            const bool stopped = step_right(g, stats, sp, b, cr, h, rmin, lane);
            dr = stopped || cr >= b;
            if (dr && !((h - rmin) >= min_prom)) return false;
        }
    }
    prom = h - fmaxf(lmin, rmin);
    return prom >= min_prom;
}

// ---------------------------------------------------------------------------
// One piece of a chunk (a full 1024-score tile, or the raw head / tail piece): local
// maxima with flat tops whose height can qualify, then their prominence.
//
// The piece and a halo of kHalo scores on either side are staged in LDS first (as
// score_for_min sees them: raw where written, the run's exact minimum elsewhere), and with
// them the (min, max) of every 32-score run of the window.  A candidate is then settled as
// cheaply as possible, in stages of growing cost:
//   0. the run summaries: a strictly higher score in the candidate's own run or in one of the
//      next runs, with the minimum over the runs passed on the way less than min_prom below the
//      candidate, rejects it after one to three 8-byte LDS reads (for a score array whose
//      wiggles are smaller than min_prom that is every candidate that is not its run's maximum,
//      and most that are);
//   1. its own thread looks kNear scores to either side.  A score array that is not white
//      has a local maximum every few scores (noise on top of whatever moves slowly), and for
//      almost all of them a strictly higher score lies a few positions away, before the
//      running minimum has dropped by min_prom: prominence = h - max(lmin, rmin) < min_prom,
//      rejected after a handful of LDS reads;
//   2. what survives (the top of each ripple crest, say) is looked at by a whole wavefront,
//      64 scores of the LDS window per step: rejected as above, or -- both sides settled
//      inside the window by a higher score or the chunk edge -- accepted with its exact
//      prominence;
//   3. only a candidate that is the highest score of its whole window on some side goes on the
//      wave-cooperative walk through global memory (prominence()).
// All three use the walk's own arithmetic (the same minima over the same scores, the same
// comparisons), so the outcome is bit-identical to walking every candidate.
#ifndef AM_PEAK_HALO
#define AM_PEAK_HALO 256
#endif
constexpr int kHalo = AM_PEAK_HALO;
constexpr int kNear = 32;
constexpr int kWin = kTile + 2 * kHalo;
constexpr int kWinRuns = kWin / 32 + 1;   // 32-score runs (aligned to absolute multiples of 32) a window can touch

struct ChunkView {
    const float* g;
    const float2* stats;
    SparseScores sp;
    long long a, b;
    float seg_min, min_prom;
    // PeakPolicy::order == 1 (distance filter first): a maximum that passes the height test but fails the
    // prominence test still takes part in the distance filter, so it is emitted as well -- with kFailedProm
    // in place of its prominence -- and removed after that filter (finish_chunk, peaks_big_finish).
    int keep_failed;
};
// the prominence of a maximum that failed the prominence test (compares false with everything)
__device__ __forceinline__ float failed_prom() { return __uint_as_float(0x7FC00000u); }
// the position the distance filter measures between (PeakPolicy::from_start)
__device__ __forceinline__ long long dist_pos(const am_peak& pk, int from_start) {
    return from_start ? (long long)pk.start : (long long)((pk.start + pk.end) / 2);
}
// dropped when closer than min_dist to a kept peak: strictly (default) or inclusively (PeakPolicy::inclusive)
__device__ __forceinline__ bool too_close(long long d, long long min_dist, int inclusive) {
    return inclusive ? d <= min_dist : d < min_dist;
}

// Scan window offsets from `from` in direction STEP until (exclusive) `end`; returns true when a
// strictly higher score was met (mn = minimum of the scores before it).  Eight scores are
// fetched from LDS per trip and then looked at in order: a one-score-per-trip loop would pay the
// LDS latency once per score.
template <int STEP>
__device__ __forceinline__ bool side_scan(const float* win, int from, int end, float h, float& mn) {
    int j = from;
    while ((end - j) * STEP >= 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = win[j + q * STEP];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (v[q] > h) return true;
            mn = fminf(mn, v[q]);
        }
        j += 8 * STEP;
    }
    for (; j != end; j += STEP) {
        const float v = win[j];
        if (v > h) return true;
        mn = fminf(mn, v);
    }
    return false;
}

// The same by a whole wavefront, 64 window offsets per step (all lanes take part).
template <int STEP>
__device__ __forceinline__ bool side_scan_wave(const float* win, int from, int end, float h, float& mn, int lane) {
    for (int j = from; (end - j) * STEP > 0; j += 64 * STEP) {
        const int idx = j + lane * STEP;
        const bool valid = (end - idx) * STEP > 0;
        const float v = valid ? win[idx] : 0.0f;
        const unsigned long long higher = __ballot(valid && v > h);
        const int nval = __popcll(__ballot(valid));
        const int ntake = higher ? (__ffsll((long long)higher) - 1) : nval;
        mn = fminf(mn, wave_min(lane < ntake ? v : FLT_MAX));
        if (higher) return true;
    }
    return false;
}

template <class Emit>
__device__ void scan_piece(const ChunkView& cv, long long lo, long long hi, float* win, float2* wruns, Cand* queue, int qcap,
                           int* queue_n, int* overflow, int tid, Emit emit) {
    const int lane = tid & 63, wv = tid >> 6;
    const long long a = cv.a, b = cv.b;
    // window [w_lo, w_hi) = piece + halo, clipped to the chunk
    const long long w_lo = lo - kHalo > a ? lo - kHalo : a;
    const long long w_hi = hi + kHalo < b ? hi + kHalo : b;
    const int wn = (int)(w_hi - w_lo);
    // staging: which runs of the window were written is looked up once per run (K3 writes whole
    // tiles, so a 32-score run is written or not as a whole); the scores then come in as
    // independent loads, kWin / 256 per thread, with the unwritten runs filled from their minimum
    const long long rb0 = w_lo >> 5;
    const int nr = (int)(((w_hi - 1) >> 5) - rb0) + 1;
    if (tid < nr) {
        const long long first = ((rb0 + tid) << 5) > w_lo ? ((rb0 + tid) << 5) : w_lo;
        const bool wr = run_written(cv.sp, first);
        wruns[tid] = make_float2(wr ? 1.0f : 0.0f, wr ? 0.0f : cv.sp.stats32[first >> 5].x);
    }
    __syncthreads();
    {
        float v[kWin / kPeakThreads];
#pragma unroll
        for (int q = 0; q < kWin / kPeakThreads; ++q) {
            const long long i = w_lo + tid + q * kPeakThreads;
            v[q] = (i < w_hi && wruns[(i >> 5) - rb0].x != 0.0f) ? cv.g[i] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < kWin / kPeakThreads; ++q) {
            const long long i = w_lo + tid + q * kPeakThreads;
            if (i < w_hi) { const float2 f = wruns[(i >> 5) - rb0]; win[i - w_lo] = f.x != 0.0f ? v[q] : f.y; }
        }
    }
    __syncthreads();
    // (min, max) of every 32-score run the window touches (over the part inside the window): eight
    // lanes per run, four scores each
    for (int r = tid >> 3; r < nr; r += kPeakThreads >> 3) {
        const long long base = ((rb0 + r) << 5) + (tid & 7) * 4;
        float mn = FLT_MAX, mx = -FLT_MAX;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (base + q >= w_lo && base + q < w_hi) { const float v = win[base + q - w_lo]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
        if ((tid & 7) == 0) wruns[r] = make_float2(mn, mx);
    }
    __syncthreads();
    for (long long i = lo + tid; i < hi; i += kPeakThreads) {
        if (i <= a || i >= b - 1) continue;
        const int wi = (int)(i - w_lo);
        const float x = win[wi];
        // (an unwritten score shows its run's minimum here, which is below theta and therefore
        // fails this test whenever the chunk's certificate holds)
        if (!((x - cv.seg_min) >= cv.min_prom) || !(win[wi - 1] < x)) continue;
        long long k = i + 1;
        while (k < b - 1 && (k < w_hi ? win[k - w_lo] : score_for_cmp(cv.g, cv.sp, k)) == x) ++k;
        if (!((k < w_hi ? win[k - w_lo] : score_for_cmp(cv.g, cv.sp, k)) < x)) continue;
        // stage 0: the run summaries.  A strictly higher score that can be reached before the scores
        // have dropped by min_prom settles the candidate against it, and both facts can be read off
        // whole runs: run maxima say where a higher score is, and the minimum over the runs passed
        // on the way (the candidate's own run included) bounds the minimum of the path from below.
        // Rejections only; everything else goes on to the exact stages.
        {
            const int R = (int)((i >> 5) - rb0);
            const float2 own = wruns[R];
            bool rejected = false;
            if (!((x - own.x) >= cv.min_prom)) {
                rejected = own.y > x;
                float acc = own.x;
                for (int d = 1; !rejected && d <= 8 && R + d < nr; ++d) {
                    const float2 st = wruns[R + d];
                    acc = fminf(acc, st.x);
                    if ((x - acc) >= cv.min_prom) break;
                    rejected = st.y > x;
                }
                acc = own.x;
                for (int d = 1; !rejected && d <= 8 && R - d >= 0; ++d) {
                    const float2 st = wruns[R - d];
                    acc = fminf(acc, st.x);
                    if ((x - acc) >= cv.min_prom) break;
                    rejected = st.y > x;
                }
            }
            if (rejected) { if (cv.keep_failed) emit((long long)i, k, x, failed_prom()); continue; }
        }
        // stage 1: flat-topped maximum [i, k) of height x against its kNear neighbours on either side
        // (a side is settled by a strictly higher score, or by the chunk edge)
        if (k <= w_hi) {
            const int wk = (int)(k - w_lo);
            const int l_end = wi - 1 - kNear > -1 ? wi - 1 - kNear : -1, r_end = wk + kNear < wn ? wk + kNear : wn;
            float lmn = x, rmn = x;
            const bool dl = side_scan<-1>(win, wi - 1, l_end, x, lmn) || (l_end == -1 && w_lo == a);
            if (dl && !((x - lmn) >= cv.min_prom)) { if (cv.keep_failed) emit((long long)i, k, x, failed_prom()); continue; }
            const bool dr = side_scan<1>(win, wk, r_end, x, rmn) || (r_end == wn && w_hi == b);
            if (dr && !((x - rmn) >= cv.min_prom)) { if (cv.keep_failed) emit((long long)i, k, x, failed_prom()); continue; }
            if (dl && dr) {
                emit((long long)i, k, x, x - fmaxf(lmn, rmn));
                continue;
            }
        }
        const int slot = atomicAdd(queue_n, 1);
        if (slot < qcap) { queue[slot].ps = i; queue[slot].pe = k; queue[slot].h = x; }
        else *overflow = 1;
    }
    __syncthreads();
    const int qn = *queue_n < qcap ? *queue_n : qcap;
    for (int q = wv; q < qn; q += kWaves) {
        const Cand cd = queue[q];
        // stage 2: the whole window, one wavefront per candidate
        if (cd.pe <= w_hi) {
            float lmn = cd.h, rmn = cd.h;
            const bool dl = side_scan_wave<-1>(win, (int)(cd.ps - 1 - w_lo), -1, cd.h, lmn, lane) || w_lo == a;
            if (dl && !((cd.h - lmn) >= cv.min_prom)) { if (cv.keep_failed && lane == 0) emit(cd.ps, cd.pe, cd.h, failed_prom()); continue; }
            const bool dr = side_scan_wave<1>(win, (int)(cd.pe - w_lo), wn, cd.h, rmn, lane) || w_hi == b;
            if (dr && !((cd.h - rmn) >= cv.min_prom)) { if (cv.keep_failed && lane == 0) emit(cd.ps, cd.pe, cd.h, failed_prom()); continue; }
            if (dl && dr) {
                if (lane == 0) emit(cd.ps, cd.pe, cd.h, cd.h - fmaxf(lmn, rmn));
                continue;
            }
        }
        // stage 3: the walk through global memory
        float prom = 0.0f;
        const bool keep = prominence(cv.g, cv.stats, cv.sp, a, b, cd.ps, cd.pe, cd.h, cv.min_prom, lane, prom);
        if (lane == 0 && (keep || cv.keep_failed)) emit(cd.ps, cd.pe, cd.h, keep ? prom : failed_prom());
    }
    __syncthreads();
    if (tid == 0) *queue_n = 0;
    __syncthreads();
}

// The end of find_peaks for one chunk: `rn` peaks that passed the prominence filter sit in
// res[]; order them by height descending (ties: position ascending), apply the greedy
// min_distance filter, write the result header, the chunk's output list and, for lists
// longer than a header holds, the spill arena.
// (with PeakPolicy::order == 1 the list also holds the maxima that failed the prominence test: they take part
// in the distance filter -- a kept one suppresses its lower neighbours -- and are left out of the result)
__device__ void finish_chunk(am_peak* res, int rn, int* order, int overflow, long long min_dist, float seg_min,
                             am_peak* my_out, SegHeader* hd, const PeakArena& arena, int* kept_s, int* spill_off_s, int tid,
                             const PeakPolicy& pol) {
    for (int i = tid; i < rn; i += kPeakThreads) {
        const float hi_ = res[i].height; const uint64_t si = res[i].start;
        int rank = 0;
        for (int j = 0; j < rn; ++j) {
            const float hj = res[j].height;
            if (hj > hi_ || (hj == hi_ && res[j].start < si)) ++rank;
        }
        order[rank] = i;
    }
    __syncthreads();
    // ---- min_distance: greedy by descending height (serial, rn is small) ----
    if (tid == 0) {
        int kept = 0, nfilt = 0;   // peaks in the result; peaks the distance filter kept (order[0 .. nfilt): nfilt <= r, so the slots are free)
        for (int r = 0; r < rn; ++r) {
            const int ir = order[r];
            const am_peak pk = res[ir];
            const long long mid = dist_pos(pk, pol.from_start);
            bool ok = true;
            if (min_dist > 0) {
                for (int k = 0; k < nfilt && ok; ++k) {
                    const long long mk = dist_pos(res[order[k]], pol.from_start);
                    const long long d = mid > mk ? mid - mk : mk - mid;
                    if (too_close(d, min_dist, pol.inclusive)) ok = false;
                }
            }
            if (!ok) continue;
            order[nfilt++] = ir;
            if (pol.order && !(pk.prominence == pk.prominence)) continue;   // failed the prominence test (failed_prom)
            if (kept < kInlinePeaks) hd->first[kept] = pk;
            my_out[kept++] = pk;
        }
        // a list longer than the header holds goes to the spill arena as a whole
        int off = -1, ovf = overflow;
        if (kept > kInlinePeaks && arena.base != nullptr) {
            const unsigned o = atomicAdd(arena.cursor, (unsigned)kept);
            if (o + (unsigned)kept <= arena.cap) off = (int)o;
            else ovf |= 4;
        }
        *kept_s = kept; *spill_off_s = off;
        hd->n = kept;
        hd->overflow = ovf;
        hd->seg_min = seg_min;
        hd->arena_off = off;
    }
    __syncthreads();
    if (*spill_off_s >= 0) {
        __threadfence_block();   // thread 0's my_out stores are visible to the block (same CU)
        for (int i = tid; i < *kept_s; i += kPeakThreads) arena.base[*spill_off_s + i] = my_out[i];
    }
}

// min_distance >= chunk length (the reference's default): the distance filter keeps exactly the
// first peak in (height descending, position ascending) order that passed the prominence filter,
// so the general path does not have to list the peaks that pass (there can be tens of thousands
// in a chunk whose scores ripple by more than min_prom) -- a running maximum over an order-
// preserving key is enough, and nothing can overflow.  The winner's plateau end and prominence
// are worked out again at the end (one scan, one walk).
__device__ __forceinline__ unsigned long long best_key(float h, long long rel) {
    unsigned u = __float_as_uint(h + 0.0f);   // (-0.0 and +0.0 are the same height: one key)
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)rel);
}
__device__ void finish_best(const ChunkView& cv, unsigned long long key, am_peak* my_out, SegHeader* hd,
                            long long* pe_s, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    if (key == 0ull) {
        if (tid == 0) { hd->n = 0; hd->overflow = 0; hd->seg_min = cv.seg_min; hd->arena_off = -1; }
        return;
    }
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    const long long ps = cv.a + (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    // the height as the score array holds it (the key does not tell -0.0 from +0.0); a peak that
    // passed the prominence filter lies in a written tile
    const float hk = __uint_as_float(u), hs = score_for_cmp(cv.g, cv.sp, ps);
    const float h = hs == hk ? hs : hk;
    if (tid == 0) {
        long long k = ps + 1;
        while (k < cv.b - 1 && score_for_cmp(cv.g, cv.sp, k) == h) ++k;
        *pe_s = k;
    }
    __syncthreads();
    if (wv == 0) {
        float prom = 0.0f;
        const bool keep = prominence(cv.g, cv.stats, cv.sp, cv.a, cv.b, ps, *pe_s, h, cv.min_prom, lane, prom);
        if (lane == 0) {
            if (keep) {
                am_peak pk; pk.start = (uint64_t)ps; pk.end = (uint64_t)*pe_s; pk.height = h; pk.prominence = prom;
                hd->first[0] = pk;
                my_out[0] = pk;
            }
            hd->n = keep ? 1 : 0; hd->overflow = 0; hd->seg_min = cv.seg_min; hd->arena_off = -1;
        }
    }
}

// Chunks whose general path would visit more than kWideTiles candidate tiles are not
// finished by their one workgroup: peaks_kernel marks them in `wide` and the two kernels
// behind it take over -- peaks_wide spreads a chunk's pieces over kWideParts workgroups that
// append the peaks passing the prominence filter to the chunk's list, peaks_finish sorts and
// filters that list.  (A score array that is not white -- speech or music against a jingle --
// has thousands of candidate maxima per chunk; one workgroup per chunk would leave 196 of
// the 256 CUs idle for tens of milliseconds.)
constexpr int kWideTiles = 0;
constexpr int kWideParts = 64;
constexpr int kWideRows = 16;    // chunks the grid of peaks_wide covers at a time (it strides over the rest)
// a piece of at most kTile scores has at most kTile / 2 flat-topped maxima: the queue cannot overflow
constexpr int kWideQueue = kTile / 2;

// Can a peak inside full tile t reach min_prom at all?  From the tile summaries alone: a peak p in t
// has height h <= M (the tile's maximum); its walk to one side ends at the first sample above h, at
// the latest inside the nearest tile u on that side whose maximum exceeds M, so the minimum over its
// walk is at least L = min of the tile minima from t to u, and
//   prominence(p) = fl(h - max(left_min, right_min)) <= fl(M - L)       (fl is monotone).
// One side with a stop tile and fl(M - L) < min_prom therefore settles the whole tile without a
// look at its scores.  A score array that drifts slowly under a ripple smaller than min_prom -- what
// a tonal or band-limited signal gives -- is rejected tile by tile this way: the uphill neighbour
// is the stop tile.  Inconclusive (true) when neither side settles within kBoundSteps tiles or the
// full tiles [tf, tl) of the chunk end first.
constexpr int kBoundSteps = 32;
// `tiles[i]` is the summary of tile tf + i (the chunk's full tiles, in LDS or in global memory).
__device__ __forceinline__ bool tile_can_qualify(const float2* tiles, long long t, long long tf, long long tl, float min_prom) {
    const float2* stats = tiles - tf;
    const float2 own = stats[t];
    const float M = own.y;
    float L = own.x;
    if ((M - L) >= min_prom) return true;                 // the tile's own range allows it
    for (long long u = t - 1, n = 0; u >= tf && n < kBoundSteps; --u, ++n) {
        const float2 v = stats[u];
        L = fminf(L, v.x);
        if ((M - L) >= min_prom) break;                   // this side cannot settle it any more
        if (v.y > M) return false;
    }
    float R = own.x;
    for (long long u = t + 1, n = 0; u < tl && n < kBoundSteps; ++u, ++n) {
        const float2 v = stats[u];
        R = fminf(R, v.x);
        if ((M - R) >= min_prom) break;
        if (v.y > M) return false;
    }
    return true;
}

// One launch for the needles of a group (several needles against one haystack: the same chunks, one score array,
// summary and flag set per needle): needle z sees its own arrays through the pointers the kernels index with the
// chunk number -- results, hand-over state and lists are laid out needle after needle.
__device__ __forceinline__ void pick_group_view(const PickGroup& grp, unsigned z, unsigned nsegs, const float*& g, const float2*& stats,
                                                SparseScores& sp, SegHeader*& hdr, am_peak*& out, WideState& wide) {
    g = grp.g[z];
    stats = grp.stats[z];
    sp.stats32 = grp.stats32[z]; sp.wbits = grp.wbits[z]; sp.tile_theta = grp.theta[z];
    hdr += grp.hdr_off[z];
    const size_t zs = (size_t)z * nsegs;
    if (out != nullptr) out += zs * AM_MAX_PEAKS_PER_CHUNK;
    wide.state += zs; wide.count += zs; wide.seg_min += zs; wide.best += zs; wide.ntiles += zs;
    if (wide.tiles != nullptr) wide.tiles += zs * kWideTileList;
    if (wide.list != nullptr) wide.list += zs * wide.cap;
}

__global__ void __launch_bounds__(kPeakThreads)
peaks_kernel(const float* g, long long g_len, const float2* stats,
             const Segment* __restrict__ segs, float min_prom, long long min_dist,
             am_peak* out, SegHeader* hdr, SparseScores sp, PeakArena arena, WideState wide,
             int only_failed, PeakPolicy pol, PickGroup grp) {
    __shared__ float red[kWaves];
    __shared__ float seg_min_s;
    __shared__ Cand queue[kQueueCap];
    __shared__ int queue_n;
    __shared__ am_peak res[AM_MAX_PEAKS_PER_CHUNK];
    __shared__ int res_n;
    __shared__ int order[AM_MAX_PEAKS_PER_CHUNK];
    __shared__ int overflow;
    __shared__ int cand_tiles[kCandCap];
    __shared__ int cand_n;
    __shared__ int kept_s, spill_off_s;
    __shared__ float win[kWin];
    __shared__ float2 wruns[kWinRuns];
    __shared__ unsigned long long best_s;
    __shared__ long long pe_s;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (grp.n > 0) pick_group_view(grp, blockIdx.y, gridDim.x, g, stats, sp, hdr, out, wide);
    const Segment sg = segs[blockIdx.x];
    const long long a = sg.a, b = sg.b < g_len ? sg.b : g_len;
    am_peak* my_out = out + (size_t)blockIdx.x * AM_MAX_PEAKS_PER_CHUNK;
    if (tid == 0) { queue_n = 0; res_n = 0; overflow = 0; cand_n = 0; best_s = 0ull; wide.state[blockIdx.x] = 0; }
    // second pick after a device-side redo of K3: only the chunks that failed their certificate take part
    // (the others keep their header; their hand-over state is cleared so that peaks_wide / peaks_finish skip them)
    if (only_failed && !(hdr[blockIdx.x].overflow & 2)) return;
    if (b - a < 3) {
        if (tid == 0) { hdr[blockIdx.x].n = 0; hdr[blockIdx.x].overflow = 0; hdr[blockIdx.x].seg_min = 0.f; hdr[blockIdx.x].arena_off = -1; }
        return;
    }
    // full tiles [tf, tl) lie completely inside [a, b)
    const long long tf = (a + kTile - 1) / kTile;
    const long long tl = b / kTile;
    const bool has_full = tl > tf;
    const long long head_hi = has_full ? tf * kTile : b;   // raw head piece [a, head_hi)
    const long long tail_lo = has_full ? tl * kTile : b;   // raw tail piece [tail_lo, b)
    // The summaries of the chunk's full tiles go into LDS before the candidate tiles are listed, in
    // the space of the result list (not in use before the first peak is emitted): the per-tile bound
    // walks read them many times, one dependent load per step.
    float2* tstats = reinterpret_cast<float2*>(res);
    constexpr long long kStage = (long long)(sizeof(res) / (sizeof(float2)));
    const long long nfull = has_full ? tl - tf : 0;
    const bool staged = nfull <= kStage;

    // ---- chunk minimum ----------------------------------------------------
    float mn = FLT_MAX;
    for (long long i = a + tid; i < head_hi; i += kPeakThreads) mn = fminf(mn, score_for_min(g, sp, i));
    for (long long i = tail_lo + tid; i < b; i += kPeakThreads) mn = fminf(mn, score_for_min(g, sp, i));
    if (has_full) for (long long t = tf + tid; t < tl; t += kPeakThreads) mn = fminf(mn, stats[t].x);
    mn = wave_min(mn);
    if (lane == 0) red[wv] = mn;
    __syncthreads();
    if (tid == 0) seg_min_s = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    __syncthreads();
    const float seg_min = seg_min_s;
    // Sparse raw scores are sufficient only if every score that can qualify (x - seg_min >= min_prom)
    // was written.  A run is written when its maximum reaches the threshold of its K3 tile (tile
    // minimum in the block + half a prominence, am_fft.hip): every threshold used for a block that
    // overlaps this chunk must therefore lie less than min_prom above the chunk's minimum (then
    // every candidate and everything that can stop a prominence walk was written).  Otherwise
    // report it and let the host redo this chunk with everything written.
    if (sp.wbits != nullptr) {
        __shared__ float th_s[kWaves];
        const int tiles = 1 << (sp.log_n2 - kColsLog);
        const long long b0 = a / sp.hop, b1 = (b - 1) / sp.hop;
        float th = -FLT_MAX;
        for (long long i = b0 * tiles + tid; i < (b1 + 1) * tiles; i += kPeakThreads) th = fmaxf(th, sp.tile_theta[i]);
        th = wave_max(th);
        if (lane == 0) th_s[wv] = th;
        __syncthreads();
        th = fmaxf(fmaxf(th_s[0], th_s[1]), fmaxf(th_s[2], th_s[3]));
        if (!((th - seg_min) < min_prom)) {
            if (tid == 0) { hdr[blockIdx.x].n = 0; hdr[blockIdx.x].overflow = 2; hdr[blockIdx.x].seg_min = seg_min; hdr[blockIdx.x].arena_off = -1; }
            // the block pairs that feed this chunk: K3 runs for them once more with every run written
            if (sp.redo_pairs != nullptr)
                for (long long q = (b0 >> 1) + tid; q <= (b1 >> 1); q += kPeakThreads) sp.redo_pairs[q] = 1;
            if (tid == 0 && sp.fail_flags != nullptr) sp.fail_flags[blockIdx.x] = 1;
            return;
        }
    }

    // ---- the chunk's maximum first -------------------------------------------
    // (1) No score can qualify unless max - chunk_min >= min_prom (prominence <= height -
    //     chunk_min): an early exit for every chunk without a hit.
    // (2) With min_distance >= chunk length (the reference's default: 480 s against 60 s
    //     chunks, audio_matcher.rs:228) the greedy distance filter keeps exactly the first
    //     peak in (height descending, position ascending) order that passes the prominence
    //     test and drops all others.  The chunk's maximum -- at its first position -- is
    //     first in that order, so if it is a peak and passes, it is the chunk's whole answer
    //     after ONE prominence evaluation, however many other maxima would qualify.  Anything else
    //     (maximum at a chunk edge, a plateau reaching the edge, insufficient prominence)
    //     falls through to the general path below.
    {
        __shared__ float bestv_s[kWaves];
        __shared__ long long bestp_s[kWaves];
        __shared__ long long peak_i_s, peak_k_s;
        __shared__ int fast_s;
        float bv = -FLT_MAX;
        long long bp = 0x7fffffffffffffffll;
        auto take = [&](float v, long long pos) { if (v > bv || (v == bv && pos < bp)) { bv = v; bp = pos; } };
        for (long long i = a + tid; i < head_hi; i += kPeakThreads) take(score_for_cmp(g, sp, i), i);
        for (long long i = tail_lo + tid; i < b; i += kPeakThreads) take(score_for_cmp(g, sp, i), i);
        if (has_full) for (long long t = tf + tid; t < tl; t += kPeakThreads) take(stats[t].y, t * kTile);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int olo = __shfl_xor((int)(bp & 0xffffffffll), o), ohi = __shfl_xor((int)(bp >> 32), o);
            take(ov, ((long long)ohi << 32) | (unsigned)olo);
        }
        if (lane == 0) { bestv_s[wv] = bv; bestp_s[wv] = bp; }
        if (tid == 0) { peak_i_s = 0x7fffffffffffffffll; fast_s = 0; }
        __syncthreads();
        bv = bestv_s[0]; bp = bestp_s[0];
        for (int k = 1; k < kWaves; ++k) take(bestv_s[k], bestp_s[k]);
        const float M = bv;
        if (!((M - seg_min) >= min_prom)) {
            if (tid == 0) { hdr[blockIdx.x].n = 0; hdr[blockIdx.x].overflow = 0; hdr[blockIdx.x].seg_min = seg_min; hdr[blockIdx.x].arena_off = -1; }
            return;
        }
        if (min_dist >= b - a) {
            // first position of the maximum: exact for a raw piece, else inside the winning tile
            const bool in_tile = has_full && bp >= tf * kTile && bp < tl * kTile;
            if (in_tile) {
                long long first = 0x7fffffffffffffffll;
                for (int q = tid; q < kTile; q += kPeakThreads)
                    if (score_for_cmp(g, sp, bp + q) == M && bp + q < first) first = bp + q;
                if (first != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&peak_i_s, (unsigned long long)first);
            } else if (tid == 0) peak_i_s = bp;
            __syncthreads();
            const long long pi = peak_i_s;
            if (tid == 0) {
                // flat top: x[i-1] < x[i] holds because i is the first position of the chunk maximum
                long long k = pi + 1;
                bool ok = pi > a && pi < b - 1;
                if (ok) {
                    while (k < b - 1 && score_for_cmp(g, sp, k) == M) ++k;
                    ok = score_for_cmp(g, sp, k) < M;
                }
                peak_k_s = k;
                fast_s = ok ? 1 : 0;
            }
            __syncthreads();
            if (fast_s) {
                // Nothing in the chunk is higher than its maximum, so both walks run to the chunk edges:
                // prominence = M - max(min[a, pi), min[plateau end, b)), as two reductions by the whole
                // workgroup (raw scores up to the first and from the last full tile of either range, tile
                // summaries in between) instead of a walk by one wavefront.  A minimum does not depend on
                // the order it is taken in: the same bits as prominence().
                __shared__ float side_s[2][kWaves];
                const long long pk_end = peak_k_s;
                float lm = M, rm = M;
                {
                    auto range_min = [&](long long lo, long long hi) {
                        float m = FLT_MAX;
                        long long t0 = (lo + kTile - 1) / kTile, t1 = hi / kTile;   // whole tiles inside [lo, hi)
                        if (t0 < tf) t0 = tf;
                        if (t1 > tl) t1 = tl;
                        if (!has_full || t0 >= t1) {
                            for (long long i = lo + tid; i < hi; i += kPeakThreads) m = fminf(m, score_for_min(g, sp, i));
                            return m;
                        }
                        for (long long i = lo + tid; i < t0 * kTile; i += kPeakThreads) m = fminf(m, score_for_min(g, sp, i));
                        for (long long t = t0 + tid; t < t1; t += kPeakThreads) m = fminf(m, stats[t].x);
                        for (long long i = t1 * kTile + tid; i < hi; i += kPeakThreads) m = fminf(m, score_for_min(g, sp, i));
                        return m;
                    };
                    const float l = wave_min(range_min(a, pi)), r = wave_min(range_min(pk_end, b));
                    if (lane == 0) { side_s[0][wv] = l; side_s[1][wv] = r; }
                }
                __syncthreads();
                if (wv == 0) {
                    for (int k = 0; k < kWaves; ++k) { lm = fminf(lm, side_s[0][k]); rm = fminf(rm, side_s[1][k]); }
                    const float prom = M - fmaxf(lm, rm);
                    const bool keep = prom >= min_prom;
                    if (lane == 0) {
                        if (keep) {
                            am_peak pk; pk.start = (uint64_t)pi; pk.end = (uint64_t)peak_k_s; pk.height = M; pk.prominence = prom;
                            hdr[blockIdx.x].first[0] = pk;
                            my_out[0] = pk;
                            hdr[blockIdx.x].n = 1; hdr[blockIdx.x].overflow = 0; hdr[blockIdx.x].seg_min = seg_min; hdr[blockIdx.x].arena_off = -1;
                            fast_s = 2;
                        } else if (pol.order) {
                            // distance filter first: the chunk's highest maximum is its only survivor, and it
                            // has just failed the prominence test -- the chunk has no peak
                            hdr[blockIdx.x].n = 0; hdr[blockIdx.x].overflow = 0; hdr[blockIdx.x].seg_min = seg_min; hdr[blockIdx.x].arena_off = -1;
                            fast_s = 2;
                        } else fast_s = 0;
                    }
                }
                __syncthreads();
                if (fast_s == 2) return;
            }
        }
    }

    // (only chunks that get this far pay for the staging: a chunk without a hit left above)
    if (staged) for (long long i = tid; i < nfull; i += kPeakThreads) tstats[i] = stats[tf + i];
    __syncthreads();
    // ---- candidate tiles: prominence <= height - chunk_min (monotone f32
    // rounding), so a tile whose maximum fails the test cannot hold a peak ----
    if (has_full) {
        for (long long t = tf + tid; t < tl; t += kPeakThreads) {
            if (!((stats[t].y - seg_min) >= min_prom)) continue;
            // (the tile-level prominence bound rejects maxima the distance-first order still needs)
            if (pol.order || (staged ? tile_can_qualify(tstats, t, tf, tl, min_prom) : tile_can_qualify(stats + tf, t, tf, tl, min_prom))) {
                const int slot = atomicAdd(&cand_n, 1);
                if (slot < kCandCap) cand_tiles[slot] = (int)(t - tf);
            }
        }
    }
    // ---- the raw head / tail piece against the full tiles inwards of it: the same bound (the piece's
    // maximum, the minimum over the piece and the tiles up to the first one with a higher maximum)
    __shared__ float pmm_s[4][kWaves];
    __shared__ int piece_can_s;
    {
        float hmn = FLT_MAX, hmx = -FLT_MAX, tmn = FLT_MAX, tmx = -FLT_MAX;
        if (has_full) {
            for (long long i = a + tid; i < head_hi; i += kPeakThreads) {
                hmn = fminf(hmn, score_for_min(g, sp, i)); hmx = fmaxf(hmx, score_for_cmp(g, sp, i));
            }
            for (long long i = tail_lo + tid; i < b; i += kPeakThreads) {
                tmn = fminf(tmn, score_for_min(g, sp, i)); tmx = fmaxf(tmx, score_for_cmp(g, sp, i));
            }
        }
        hmn = wave_min(hmn); hmx = wave_max(hmx); tmn = wave_min(tmn); tmx = wave_max(tmx);
        if (lane == 0) { pmm_s[0][wv] = hmn; pmm_s[1][wv] = hmx; pmm_s[2][wv] = tmn; pmm_s[3][wv] = tmx; }
    }
    __syncthreads();
    if (tid == 0) {
        int can = (head_hi > a ? 1 : 0) | (b > tail_lo ? 2 : 0);
        if (has_full && !pol.order) {
            const float2* tl_stats = staged ? tstats : stats + tf;
            float m = FLT_MAX, M = -FLT_MAX;
            for (int k = 0; k < kWaves; ++k) { m = fminf(m, pmm_s[0][k]); M = fmaxf(M, pmm_s[1][k]); }
            if ((can & 1) && !((M - m) >= min_prom)) {
                float L = m;
                for (long long u = 0; u < nfull && u < kBoundSteps; ++u) {
                    const float2 v = tl_stats[u];
                    L = fminf(L, v.x);
                    if ((M - L) >= min_prom) break;
                    if (v.y > M) { can &= ~1; break; }
                }
            }
            m = FLT_MAX; M = -FLT_MAX;
            for (int k = 0; k < kWaves; ++k) { m = fminf(m, pmm_s[2][k]); M = fmaxf(M, pmm_s[3][k]); }
            if ((can & 2) && !((M - m) >= min_prom)) {
                float L = m;
                for (long long u = nfull - 1; u >= 0 && u >= nfull - kBoundSteps; --u) {
                    const float2 v = tl_stats[u];
                    L = fminf(L, v.x);
                    if ((M - L) >= min_prom) break;
                    if (v.y > M) { can &= ~2; break; }
                }
            }
        }
        piece_can_s = can;
    }
    __syncthreads();
    const int piece_can = piece_can_s;
    if (cand_n == 0 && piece_can == 0) {   // nothing in this chunk can qualify
        if (tid == 0) { hdr[blockIdx.x].n = 0; hdr[blockIdx.x].overflow = 0; hdr[blockIdx.x].seg_min = seg_min; hdr[blockIdx.x].arena_off = -1; }
        return;
    }
    if (cand_n > kWideTiles && wide.list != nullptr) {
        // too much for one workgroup: hand the chunk to peaks_wide / peaks_finish, with the list of
        // candidate tiles if it is complete (otherwise the parts test every tile themselves)
        const bool listed = cand_n <= kCandCap;
        if (listed) for (int i = tid; i < cand_n; i += kPeakThreads) wide.tiles[(size_t)blockIdx.x * kCandCap + i] = cand_tiles[i];
        if (tid == 0) {
            wide.seg_min[blockIdx.x] = seg_min; wide.count[blockIdx.x] = 0; wide.best[blockIdx.x] = 0ull;
            wide.ntiles[blockIdx.x] = listed ? cand_n : -1;
            wide.state[blockIdx.x] = 1 | (piece_can << 1);   // bit 0: handed over; bits 1, 2: the head / tail piece needs a scan
        }
        return;
    }
    // more candidates than the list holds: visit every full tile instead
    const bool all_tiles = cand_n > kCandCap;
    const long long nmid = has_full ? (all_tiles ? (tl - tf) : cand_n) : 0;
    ChunkView cv{g, stats, sp, a, b, seg_min, min_prom, pol.order};
    const bool best_mode = min_dist >= b - a && b - a < 0xFFFFFFFFll;
    auto emit = [&](long long ps, long long pe, float h, float prom) {
        if (best_mode) { atomicMax(&best_s, best_key(h, ps - a)); return; }
        const int slot = atomicAdd(&res_n, 1);
        if (slot < AM_MAX_PEAKS_PER_CHUNK) {
            res[slot].start = (uint64_t)ps; res[slot].end = (uint64_t)pe;
            res[slot].height = h; res[slot].prominence = prom;
        } else overflow = 1;
    };
    // ---- pieces: head, candidate tiles, tail --------------------------------
    const long long npieces = nmid + 2;
    for (long long pc = 0; pc < npieces; ++pc) {
        long long lo, hi;
        if (pc == 0) { lo = a; hi = (piece_can & 1) ? head_hi : a; }
        else if (pc == npieces - 1) { lo = tail_lo; hi = (piece_can & 2) ? b : tail_lo; }
        else {
            const long long t = tf + (all_tiles ? (pc - 1) : (long long)cand_tiles[pc - 1]);
            lo = t * kTile; hi = lo + kTile;
            if (all_tiles && !((stats[t].y - seg_min) >= min_prom && (pol.order || tile_can_qualify(stats + tf, t, tf, tl, min_prom)))) continue;
        }
        // (a chunk without a full tile inside has a head piece of up to 2 * kTile - 2 scores:
        // the LDS window holds kTile + halo, so long pieces go in slices)
        for (long long q0 = lo; q0 < hi; q0 += kTile)
            scan_piece(cv, q0, q0 + kTile < hi ? q0 + kTile : hi, win, wruns, queue, kQueueCap, &queue_n, &overflow, tid, emit);
    }
    if (best_mode) {
        finish_best(cv, best_s, my_out, &hdr[blockIdx.x], &pe_s, tid);
        return;
    }
    const int rn = res_n < AM_MAX_PEAKS_PER_CHUNK ? res_n : AM_MAX_PEAKS_PER_CHUNK;
    finish_chunk(res, rn, order, overflow, min_dist, seg_min, my_out, &hdr[blockIdx.x], arena, &kept_s, &spill_off_s, tid, pol);
}

// grid (kWideParts, nsegs): part p of chunk s takes the head piece (p == 0), the tail piece
// (p == 1 mod parts) and every kWideParts-th full tile, and appends what passes the
// prominence filter to the chunk's list.
__global__ void __launch_bounds__(kPeakThreads)
peaks_wide(const float* g, long long g_len, const float2* stats,
           const Segment* __restrict__ segs, int nsegs, float min_prom, long long min_dist, SparseScores sp, WideState wide, PeakPolicy pol,
           PickGroup grp) {
    __shared__ Cand queue[kWideQueue];
    __shared__ int queue_n;
    __shared__ int overflow;
    __shared__ float win[kWin];
    __shared__ float2 wruns[kWinRuns];
    const int part = blockIdx.x, tid = threadIdx.x;
    if (grp.n > 0) {
        SegHeader* no_hdr = nullptr; am_peak* no_out = nullptr;
        pick_group_view(grp, blockIdx.z, (unsigned)nsegs, g, stats, sp, no_hdr, no_out, wide);
    }
    // The grid covers kWideRows chunks at a time and strides over the rest: almost every chunk is finished by peaks_kernel
    // itself and only has its state looked at here -- one workgroup per (part, chunk) would be thousands of workgroups that
    // start and return (0.6 ms of dispatch per launch for the 8 x 60 chunks of a needle group).
    for (int seg = blockIdx.y; seg < nsegs; seg += gridDim.y) {
    if (!(wide.state[seg] & 1)) continue;
    const Segment sg = segs[seg];
    const long long a = sg.a, b = sg.b < g_len ? sg.b : g_len;
    const long long tf = (a + kTile - 1) / kTile;
    const long long tl = b / kTile;
    const bool has_full = tl > tf;
    const long long head_hi = has_full ? tf * kTile : b;
    const long long tail_lo = has_full ? tl * kTile : b;
    const float seg_min = wide.seg_min[seg];
    if (tid == 0) { queue_n = 0; overflow = 0; }
    __syncthreads();
    ChunkView cv{g, stats, sp, a, b, seg_min, min_prom, pol.order};
    am_peak* list = wide.list + (size_t)seg * wide.cap;
    const bool best_mode = min_dist >= b - a && b - a < 0xFFFFFFFFll;
    auto emit = [&](long long ps, long long pe, float h, float prom) {
        if (best_mode) { atomicMax(&wide.best[seg], best_key(h, ps - a)); return; }
        const unsigned slot = atomicAdd(&wide.count[seg], 1u);
        if (slot < wide.cap) {
            am_peak pk; pk.start = (uint64_t)ps; pk.end = (uint64_t)pe; pk.height = h; pk.prominence = prom;
            list[slot] = pk;
        }
    };
    const int st = wide.state[seg];
    if (part == 0 && (st & 2))
        for (long long q0 = a; q0 < head_hi; q0 += kTile)
            scan_piece(cv, q0, q0 + kTile < head_hi ? q0 + kTile : head_hi, win, wruns, queue, kWideQueue, &queue_n, &overflow, tid, emit);
    if (part == 1 % kWideParts && (st & 4) && b > tail_lo) scan_piece(cv, tail_lo, b, win, wruns, queue, kWideQueue, &queue_n, &overflow, tid, emit);
    const int nlisted = wide.ntiles[seg];
    if (has_full && nlisted >= 0) {
        // the candidate tiles peaks_kernel found, dealt round-robin
        const int* tiles = wide.tiles + (size_t)seg * kCandCap;
        for (int k = part; k < nlisted; k += kWideParts) {
            const long long t = tf + tiles[k];
            scan_piece(cv, t * kTile, (t + 1) * kTile, win, wruns, queue, kWideQueue, &queue_n, &overflow, tid, emit);
        }
    } else if (has_full) {
        for (long long t = tf + part; t < tl; t += kWideParts) {
            if (!((stats[t].y - seg_min) >= min_prom && (pol.order || tile_can_qualify(stats + tf, t, tf, tl, min_prom)))) continue;
            scan_piece(cv, t * kTile, (t + 1) * kTile, win, wruns, queue, kWideQueue, &queue_n, &overflow, tid, emit);
        }
    }
    if (tid == 0 && overflow) atomicAdd(&wide.count[seg], 0x40000000u);   // (cannot happen, see kWideQueue) poisons the count
    __syncthreads();   // (the next chunk of this workgroup resets the queue)
    }
}

// ---------------------------------------------------------------------------
// More than AM_MAX_PEAKS_PER_CHUNK peaks pass the prominence filter in one chunk: the list lives in
// global memory (built by peaks_wide with a capacity of its own) and ONE workgroup finishes it.
// (1) keys: height descending, position ascending, as one order-preserving 64-bit key (best_key).
// (2) a stable LSD radix sort, 4 bits per pass, descending: every thread owns a contiguous slice,
//     counts its digits into its own LDS column, a block scan turns the counts into offsets, the
//     thread scatters its slice in order.  16 passes, ping-pong between the two halves of keys / idx.
// (3) the greedy filter (finish_chunk's rule: keep a peak unless a kept one lies closer than
//     min_dist between plateau centres), one wavefront, 64 peaks per step in priority order: kept
//     peaks are entered into a table of min_dist-wide buckets by centre -- two kept centres are at
//     least min_dist apart, so a bucket holds at most one, and a conflict can only sit in the
//     peak's own bucket or a neighbouring one; conflicts inside the step are settled lane by lane.
constexpr int kBigThreads = 256;
__global__ void __launch_bounds__(kBigThreads)
peaks_big_finish(const am_peak* __restrict__ list, unsigned n, long long a, long long min_dist,
                 unsigned long long* keys, unsigned* idx, long long* table, am_peak* __restrict__ out, unsigned* out_n, PeakPolicy pol) {
    __shared__ unsigned cnt[16 * kBigThreads];
    __shared__ unsigned part[kBigThreads];
    const int tid = threadIdx.x;
    unsigned long long* k0 = keys; unsigned long long* k1 = keys + n;
    unsigned* i0 = idx; unsigned* i1 = idx + n;
    for (unsigned i = tid; i < n; i += kBigThreads) { k0[i] = best_key(list[i].height, (long long)list[i].start - a); i0[i] = i; }
    const unsigned per = (n + kBigThreads - 1) / kBigThreads;
    const unsigned lo = (unsigned)tid * per < n ? (unsigned)tid * per : n;
    const unsigned hi = lo + per < n ? lo + per : n;
    for (int pass = 0; pass < 16; ++pass) {
        __syncthreads();   // the previous pass's (or the key loop's) global stores are visible to the workgroup
        const unsigned long long* kin = (pass & 1) ? k1 : k0;
        unsigned long long* kout = (pass & 1) ? k0 : k1;
        const unsigned* iin = (pass & 1) ? i1 : i0;
        unsigned* iout = (pass & 1) ? i0 : i1;
        const int sh = 4 * pass;
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d * kBigThreads + tid] = 0;
        for (unsigned i = lo; i < hi; ++i) ++cnt[(15 - (int)((kin[i] >> sh) & 15ull)) * kBigThreads + tid];
        __syncthreads();
        // exclusive scan of the 4096 counts in (digit, thread) order: 16 consecutive entries per thread
        unsigned sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += cnt[tid * 16 + j];
        part[tid] = sum;
        __syncthreads();
        if (tid == 0) { unsigned run = 0; for (int t = 0; t < kBigThreads; ++t) { const unsigned v = part[t]; part[t] = run; run += v; } }
        __syncthreads();
        unsigned run = part[tid];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const unsigned v = cnt[tid * 16 + j]; cnt[tid * 16 + j] = run; run += v; }
        __syncthreads();
        for (unsigned i = lo; i < hi; ++i) {
            const unsigned long long k = kin[i];
            const unsigned pos = cnt[(15 - (int)((k >> sh) & 15ull)) * kBigThreads + tid]++;
            kout[pos] = k; iout[pos] = iin[i];
        }
    }
    __syncthreads();   // 16 passes: the sorted order is back in the first halves
    if (tid >= 64) return;
    const int lane = tid;
    unsigned kept = 0;
    for (unsigned base = 0; base < n; base += 64) {
        const unsigned i = base + lane;
        bool ok = i < n;
        am_peak pk; pk.start = 0; pk.end = 0; pk.height = 0.f; pk.prominence = 0.f;
        long long mid = 0, bkt = 0;
        if (ok) {
            pk = list[i0[i]];
            mid = dist_pos(pk, pol.from_start) - a;
            if (min_dist > 0) {
                bkt = mid / min_dist;
                for (long long bb = bkt > 0 ? bkt - 1 : 0; bb <= bkt + 1 && ok; ++bb) {
                    const long long m = __hip_atomic_load(&table[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (m >= 0) { const long long d = mid > m ? mid - m : m - mid; if (too_close(d, min_dist, pol.inclusive)) ok = false; }
                }
            }
        }
        bool keep = ok;
        if (min_dist > 0) {
            keep = false;
            unsigned long long pending = __ballot(ok);
            while (pending) {
                const int L = __ffsll((long long)pending) - 1;
                const long long midL = ((long long)__shfl((int)(mid >> 32), L) << 32) | (unsigned)__shfl((int)(mid & 0xffffffffll), L);
                if (lane == L) keep = true;
                if (ok && lane > L) { const long long d = mid > midL ? mid - midL : midL - mid; if (too_close(d, min_dist, pol.inclusive)) ok = false; }
                pending = __ballot(ok && lane > L);
            }
        }
        // (distance-first order: a kept maximum that failed the prominence test enters the table, not the result)
        const bool result = keep && !(pol.order && !(pk.prominence == pk.prominence));
        const unsigned long long kb = __ballot(result);
        if (result) out[kept + (unsigned)__popcll(kb & ((1ull << lane) - 1ull))] = pk;
        if (keep && min_dist > 0) __hip_atomic_store(&table[bkt], mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        kept += (unsigned)__popcll(kb);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // this step's table entries before the next step's look-ups
    }
    if (lane == 0) *out_n = kept;
}

// grid nsegs: sort + distance filter of a chunk that went through peaks_wide
__global__ void __launch_bounds__(kPeakThreads)
peaks_finish(const float* g, long long g_len, const float2* stats,
             const Segment* __restrict__ segs, float min_prom, long long min_dist, am_peak* out,
             SegHeader* hdr, SparseScores sp, PeakArena arena, WideState wide, PeakPolicy pol, PickGroup grp) {
    __shared__ am_peak res[AM_MAX_PEAKS_PER_CHUNK];
    __shared__ int order[AM_MAX_PEAKS_PER_CHUNK];
    __shared__ int kept_s, spill_off_s;
    __shared__ long long pe_s;
    const int seg = blockIdx.x, tid = threadIdx.x;
    if (grp.n > 0) pick_group_view(grp, blockIdx.y, gridDim.x, g, stats, sp, hdr, out, wide);
    if (!(wide.state[seg] & 1)) return;
    am_peak* my_out = out + (size_t)seg * AM_MAX_PEAKS_PER_CHUNK;
    const Segment sg = segs[seg];
    const long long a = sg.a, b = sg.b < g_len ? sg.b : g_len;
    if (min_dist >= b - a && b - a < 0xFFFFFFFFll) {
        ChunkView cv{g, stats, sp, a, b, wide.seg_min[seg], min_prom, pol.order};
        finish_best(cv, wide.best[seg], my_out, &hdr[seg], &pe_s, tid);
        return;
    }
    const unsigned cnt = wide.count[seg];
    const int rn = cnt < (unsigned)AM_MAX_PEAKS_PER_CHUNK ? (int)cnt : AM_MAX_PEAKS_PER_CHUNK;
    const am_peak* list = wide.list + (size_t)seg * AM_MAX_PEAKS_PER_CHUNK;
    for (int i = tid; i < rn; i += kPeakThreads) res[i] = list[i];
    __syncthreads();
    finish_chunk(res, rn, order, cnt > (unsigned)AM_MAX_PEAKS_PER_CHUNK ? 1 : 0, min_dist, wide.seg_min[seg],
                 my_out, &hdr[seg], arena, &kept_s, &spill_off_s, tid, pol);
}

// ---------------------------------------------------------------------------
// sum of squares in f64 (CorrelateAlgo::inverse_sample_auto_correlation,
// audio_matcher.rs:321-329: element 0 of the needle's autocorrelation).  One
// partial sum per workgroup; the host adds them in index order, so the result
// does not depend on the order in which workgroups finish.
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, long long n, double* out) {
    __shared__ double part[4];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double v = (double)x[i];
        acc += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}

// counter-based test signal: 24 hashed bits -> [-1, 1) exactly, times amp.
// tests/ compare it bit for bit with the checker's own generator.
__global__ void __launch_bounds__(256) synth_kernel(float* __restrict__ out, uint32_t seed, uint32_t stream,
                                                    uint64_t first, long long n, float amp) {
    const uint32_t key = fmix32(seed * 0x9E3779B9u + stream * 0x7F4A7C15u + 0x01234567u);
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
        const uint64_t i = first + (uint64_t)k;
        uint32_t h = fmix32((uint32_t)i ^ key);
        h = fmix32(h + stream * 0x9E3779B9u + (uint32_t)(i >> 32) * 0xC2B2AE35u + seed);
        const int32_t v = (int32_t)(h >> 8) - (1 << 23);
        out[k] = ((float)v * (1.0f / 8388608.0f)) * amp;
    }
}

// the same signal as interleaved i16 stereo frames: channel L = stream,
// channel R = stream + 5000, each rint(uniform * amp * 32767) (round half to even), saturated
__device__ __forceinline__ float synth_value(uint32_t seed, uint32_t stream, uint64_t i, float amp) {
    const uint32_t key = fmix32(seed * 0x9E3779B9u + stream * 0x7F4A7C15u + 0x01234567u);
    uint32_t h = fmix32((uint32_t)i ^ key);
    h = fmix32(h + stream * 0x9E3779B9u + (uint32_t)(i >> 32) * 0xC2B2AE35u + seed);
    const int32_t v = (int32_t)(h >> 8) - (1 << 23);
    return __fmul_rn(__fmul_rn((float)v, 1.0f / 8388608.0f), amp);
}
__device__ __forceinline__ short to_s16(float x) {
    const float r = rintf(__fmul_rn(x, 32767.0f));
    return (short)fminf(fmaxf(r, -32768.0f), 32767.0f);
}
__global__ void __launch_bounds__(256) synth_pcm16_kernel(short2* __restrict__ out, uint32_t seed, uint32_t stream,
                                                          uint64_t first, long long frames, float amp) {
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < frames; k += (long long)gridDim.x * 256) {
        const uint64_t i = first + (uint64_t)k;
        out[k] = make_short2(to_s16(synth_value(seed, stream, i, amp)), to_s16(synth_value(seed, stream + 5000u, i, amp)));
    }
}
// dst[i] = saturate(dst[i] + src[i]) on interleaved i16 values (plants a needle into a haystack)
__global__ void __launch_bounds__(256) add_pcm16_kernel(short* __restrict__ dst, const short* __restrict__ src, long long n) {
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
        const int v = (int)dst[k] + (int)src[k];
        dst[k] = (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
    }
}

__global__ void __launch_bounds__(256) axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n, float gain) {
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256)
        dst[k] = __fadd_rn(dst[k], __fmul_rn(gain, src[k]));
}

// mp3_reader.rs:12, 28-37: (l as f32 + r as f32) * 0.5 * PCM_FACTOR, each step in f32
__global__ void __launch_bounds__(256) pcm_downmix_kernel(const int16_t* __restrict__ in, long long frames, float* __restrict__ out) {
    const float pcm_factor = 1.0f / 65535.0f;
    const short2* in2 = reinterpret_cast<const short2*>(in);
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < frames; k += (long long)gridDim.x * 256) {
        const short2 lr = in2[k];
        const float sum = __fadd_rn((float)lr.x, (float)lr.y);
        out[k] = __fmul_rn(__fmul_rn(sum, 0.5f), pcm_factor);
    }
}

// grid (parts, nranges): does a range of samples hold a NaN or an infinity?
__global__ void __launch_bounds__(256) nonfinite_ranges_kernel(const float* __restrict__ x, const Segment* __restrict__ ranges,
                                                               int* __restrict__ flags) {
    const Segment r = ranges[blockIdx.y];
    bool found = false;
    for (long long i = r.a + (long long)blockIdx.x * 256 + threadIdx.x; i < r.b; i += (long long)gridDim.x * 256)
        found |= !(fabsf(x[i]) <= FLT_MAX);
    if (found) flags[blockIdx.y] = 1;
}

static inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

hipError_t launch_tile_stats(hipStream_t st, const float* g, long long n, float2* stats, int* bad) {
    const long long tiles = (n + kTile - 1) / kTile;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_stats, dim3((unsigned)tiles), dim3(256), 0, st, g, n, stats, bad);
    return hipGetLastError();
}

hipError_t launch_stats_reduce(hipStream_t st, const float2* stats32, long long n, float2* stats, int* bad, const PickGroup* grp) {
    const long long n32 = (n + 31) / 32;
    const long long tiles = (n + kTile - 1) / kTile;
    if (tiles <= 0) return hipSuccess;
    const long long blocks = (tiles * 8 + 255) / 256;
    const PickGroup none{};
    hipLaunchKernelGGL(stats_reduce, dim3((unsigned)blocks, grp ? (unsigned)grp->n : 1u), dim3(256), 0, st, stats32, n32, stats, tiles, bad,
                       grp ? *grp : none);
    return hipGetLastError();
}

hipError_t launch_peaks(hipStream_t st, const float* g, long long g_len, const float2* stats,
                        const Segment* d_segs, int nsegs, float min_prom, long long min_dist,
                        am_peak* d_out, SegHeader* d_hdr, const SparseScores& sp, const PeakArena& arena,
                        const WideState& wide, bool only_failed, const PeakPolicy& pol, const PickGroup* grp) {
    if (nsegs <= 0) return hipSuccess;
    const PickGroup none{};
    const PickGroup& pg = grp ? *grp : none;
    const unsigned nz = grp ? (unsigned)grp->n : 1u;
    hipLaunchKernelGGL(peaks_kernel, dim3(nsegs, nz), dim3(kPeakThreads), 0, st, g, g_len, stats, d_segs,
                       min_prom, min_dist, d_out, d_hdr, sp, arena, wide, only_failed ? 1 : 0, pol, pg);
    if (wide.list != nullptr) {
        // both return at once for chunks that peaks_kernel finished itself (the usual case)
        hipLaunchKernelGGL(peaks_wide, dim3(kWideParts, nsegs < kWideRows ? nsegs : kWideRows, nz), dim3(kPeakThreads), 0, st, g, g_len, stats, d_segs,
                           nsegs, min_prom, min_dist, sp, wide, pol, pg);
        hipLaunchKernelGGL(peaks_finish, dim3(nsegs, nz), dim3(kPeakThreads), 0, st, g, g_len, stats, d_segs, min_prom, min_dist,
                           d_out, d_hdr, sp, arena, wide, pol, pg);
    }
    return hipGetLastError();
}

hipError_t launch_nonfinite_ranges(hipStream_t st, const float* x, const Segment* ranges, int nranges, int* flags) {
    if (nranges <= 0) return hipSuccess;
    hipLaunchKernelGGL(nonfinite_ranges_kernel, dim3(64, nranges), dim3(256), 0, st, x, ranges, flags);
    return hipGetLastError();
}

hipError_t launch_peaks_wide_one(hipStream_t st, const float* g, long long g_len, const float2* stats, const Segment* d_seg,
                                 float min_prom, long long min_dist, const SparseScores& sp, const WideState& wide, const PeakPolicy& pol) {
    hipLaunchKernelGGL(peaks_wide, dim3(kWideParts, 1), dim3(kPeakThreads), 0, st, g, g_len, stats, d_seg, 1, min_prom, min_dist, sp, wide, pol, PickGroup{});
    return hipGetLastError();
}
hipError_t launch_peaks_big_finish(hipStream_t st, const am_peak* list, unsigned n, long long a, long long min_dist,
                                   unsigned long long* keys, unsigned* idx, long long* table, am_peak* out, unsigned* out_n,
                                   const PeakPolicy& pol) {
    hipLaunchKernelGGL(peaks_big_finish, dim3(1), dim3(kBigThreads), 0, st, list, n, a, min_dist, keys, idx, table, out, out_n, pol);
    return hipGetLastError();
}

int sumsq_parts(long long n) { return grid_for(n); }

hipError_t launch_sumsq(hipStream_t st, const float* x, long long n, double* d_parts) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, d_parts);
    return hipGetLastError();
}

hipError_t launch_synth(hipStream_t st, float* out, uint32_t seed, uint32_t stream, uint64_t first,
                        long long n, float amp) {
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, seed, stream, first, n, amp);
    return hipGetLastError();
}

hipError_t launch_synth_pcm16(hipStream_t st, int16_t* out, uint32_t seed, uint32_t stream, uint64_t first, long long frames, float amp) {
    hipLaunchKernelGGL(synth_pcm16_kernel, dim3(grid_for(frames)), dim3(256), 0, st, reinterpret_cast<short2*>(out), seed, stream, first, frames, amp);
    return hipGetLastError();
}

hipError_t launch_add_pcm16(hipStream_t st, int16_t* dst, const int16_t* src, long long frames) {
    hipLaunchKernelGGL(add_pcm16_kernel, dim3(grid_for(2 * frames)), dim3(256), 0, st, dst, src, 2 * frames);
    return hipGetLastError();
}

hipError_t launch_axpy(hipStream_t st, float* dst, const float* src, long long n, float gain) {
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, st, dst, src, n, gain);
    return hipGetLastError();
}

hipError_t launch_pcm_downmix(hipStream_t st, const int16_t* in, long long frames, float* out) {
    hipLaunchKernelGGL(pcm_downmix_kernel, dim3(grid_for(frames)), dim3(256), 0, st, in, frames, out);
    return hipGetLastError();
}

}  // namespace am
