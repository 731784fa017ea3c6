// am_kernels.h -- device-side job descriptors and launcher prototypes shared by
// the HIP translation units of libaudiomatch_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/audiomatch.h"

namespace am {

// One overlap-save job: a (virtually zero-padded) input signal and the score
// array it produces.  score[j] = sum_n X[j + n - lead] * needle[n], X = 0
// outside [0, src_len).  Block b reads input [b*hop - lead, b*hop - lead + N)
// and yields scores [b*hop, b*hop + hop).  Two blocks are packed into one
// complex transform (re = block 2g, im = block 2g+1).
struct Job {
    const void* src;      // device: f32 mono samples, or interleaved i16 stereo frames (src_kind 1)
    long long src_len;    // samples / frames
    long long lead;       // virtual zeros in front of src[0]
    float* dst;           // device, out_count scores
    long long out_count;
    int hop;              // new scores per block (<= N - S + 1)
    int nblocks;          // ceil(out_count / hop)
    int first_pair;       // first pair handled by this launch (slot 0 of work)
    int src_kind;         // 0 = f32 mono, 1 = i16 stereo (down-mixed in K1, mp3_reader.rs:28-37)
};

// Factorisation N = N1 * N2 of the complex transform and its twiddle tables.
struct PlanDev {
    int logN, logN1, logN2, logLo;
    const float2* tw1;    // W_N1^k, k < N1/2 (forward sign)
    const float2* tw2;    // W_N2^k, k < N2/2
    const float2* twlo;   // W_N^m,            m < 2^logLo
    const float2* twhi;   // W_N^(m << logLo), m < 2^(logN - logLo)
    // The same two tables with every entry's FOURTH power beside it, (w.x, w.y, w^4.x, w^4.y): one 16-byte
    // fetch per level yields W_N^m and W_N^(4m), the two seeds of a twiddle chain (twiddle_chain(x, base, step, step4)).
    const float4* twlo4;
    const float4* twhi4;
    // K2's twiddle seeds per lane, ready to fetch (N2 = 8192 only): k2j[2t] = (W^(2t), W^(2t+1)), k2j[2t+1] = their
    // fourth powers, t < 256; k2c[2c] = (W^(32c), W^(32c+16)), k2c[2c+1] = their fourth powers, c < 16 (W = W_8192).
    const float4* k2j;
    const float4* k2c;
    // constant tables of the matrix-core row kernel (k2_rows_m16, N2 = 8192 only; nullptr otherwise): DFT-16 and DFT-32
    // operands in the lane layout of v_mfma_f32_16x16x32_f16 and every lane's twiddles as h2 (am_fft.hip kMf*)
    const unsigned* mf;
};

constexpr int kColsLog = 5;            // B = 32 columns per K1/K3 workgroup
constexpr int kCols = 1 << kColsLog;
constexpr int kTile = 1024;            // score tile of the min/max summary
constexpr int kFftThreads = 256;

// one reference chunk's slice of a score array (audio_matcher.rs:119-126)
struct Segment {
    long long a, b;       // [a, b) in the score array
};

// half: 0 = f32 work matrix; 1 = one __half2 per point, f32 butterflies; 2 = packed f16 butterflies too
// (K2 and K1 whole; K3's first pass -- its second pass and the score scan stay f32)
hipError_t launch_k1(hipStream_t st, const Job& job, int npairs, float2* work, const PlanDev& pl, int half = 0);
// dst == nullptr: in place; otherwise the result goes to a second work matrix
// tail: the launch belongs to a haystack's odd last block (am_api.hip, run_tail_block) -- the same row kernels under
// names of their own (tail_rows_*), so that a profile's per-kernel averages stay those of the main pass
hipError_t launch_k2(hipStream_t st, int npairs, float2* work, const float2* hc, const PlanDev& pl, float2* dst = nullptr,
                     int half = 0, float hscale = 1.0f, float pre = 1.0f, bool tail = false);
// half = 2: hc points at the __half2 form of the spectrum (launch_spectrum_to_half), hscale already in it
hipError_t launch_spectrum_to_half(hipStream_t st, const float2* hc, long long n, float scale, unsigned* out);
// option "k2_mfma" (an A/B experiment for half_pipeline = 2): the row kernel's butterflies on the matrix cores;
// it takes the spectrum conjugated and in [a'][b'][c'] order (launch_spectrum_to_half_mfma)
void set_k2_mfma(int on);
bool k2_mfma_enabled();
int k2_mfma_table_dwords();
hipError_t launch_spectrum_to_half_mfma(hipStream_t st, const float2* hc, long long n, float scale, unsigned* out);
hipError_t launch_k2_spectrum(hipStream_t st, float2* work, float2* hc_out, const PlanDev& pl);
// A group of needles sharing one forward row transform (r16 rows, f32 storage only):
// needle j multiplies with hc[j] and writes its inverse rows to dst[j].
constexpr int kMaxNeedleGroup = 8;
struct K2Group {
    const float2* hc[kMaxNeedleGroup];
    float2* dst[kMaxNeedleGroup];
    int n;
};
bool plan_k2_has_group(const PlanDev& pl);
hipError_t launch_k2_group(hipStream_t st, int npairs, const float2* work, const K2Group& grp, const PlanDev& pl);
// The score scan fused into K3 (r16 plan only).  stats32 == nullptr disables it
// (plain correlation: every score is written).
struct ScanCfg {
    float2* stats32;          // (min,max) per 32 consecutive scores, always written
    // which 32-score runs were written: one 64-bit word per (block, column tile, wavefront of the tile's
    // workgroup), bit (a << 2) | j = row a * (rows / 16) + 4 * wavefront + j of the tile
    unsigned long long* wbits;
    float* tile_theta;        // [block][column tile]: the write threshold that tile used
    float margin;             // a run's raw scores are written when its maximum >= min(the tile's minimum over its block
                              // pair, hist_min) + margin (and for runs that hold a chunk edge); margin < 0: every run
    float hist_min;           // lowest chunk minimum of the needle's recent haystacks (FLT_MAX: none)
    long long seg_c, seg_d;   // chunk geometry: runs that hold score i*seg_c or i*seg_c + seg_d are chunk edges
    double inv_c;             // 1.0 / seg_c
    // device-side redo: only the block pairs p with only_pairs[p] != 0 run (nullptr: all).  The peak pick
    // marks the pairs of the chunks whose certificate failed (SparseScores::redo_pairs), K3 then runs once
    // more for those pairs with every run written, from the work matrix it still has.
    const int* only_pairs;
    int redo_tiles;           // (set by launch_k3 for that launch: tiles in the launch, walked by a small grid)
    // Chunk edges per block pair of the launch, worked out on the host (launch_k3): for slot s of the launch the first
    // score i*seg_c and the first score i*seg_c + seg_d at or behind the start of block A (0, 1) and of block B (2, 3),
    // relative to that start, INT_MAX when out of reach.  edges_n = 0 (more pairs than kMaxEdgeSlots, or chunks
    // shorter than a block): the kernel works them out itself.
    static constexpr int kMaxEdgeSlots = 64;
    int edges_n;
    int edge_rel[kMaxEdgeSlots][4];
};
// accumulate: the scores are added to what job.dst holds (every run written; f32 work matrix only)
hipError_t launch_k3(hipStream_t st, const Job& job, int npairs, const float2* work,
                     const PlanDev& pl, float out_scale, const ScanCfg& scan, int half = 0, bool accumulate = false);
// K3 for the needles of a group in ONE launch (BASELINE configs[3]): needle z = blockIdx.y reads its own inverse rows
// and writes its own scores, summary, ballots and thresholds; block layout and chunk geometry are the group's (`job`,
// `scan`: their dst / stats32 / wbits / tile_theta / hist_min fields are replaced per needle).  f32 work matrices of the
// 512- and 256-row plans.
struct K3Group {
    int n;
    const float2* work[kMaxNeedleGroup];
    float* dst[kMaxNeedleGroup];
    float2* stats32[kMaxNeedleGroup];
    unsigned long long* wbits[kMaxNeedleGroup];
    float* tile_theta[kMaxNeedleGroup];
    float hist_min[kMaxNeedleGroup];
    float out_scale[kMaxNeedleGroup];
};
bool plan_k3_has_group(const PlanDev& pl);
// The odd last blocks of several haystacks of a batch (am_api.hip, TailPlan) as ONE launch each of K1 / K2 / K3 on the
// 256-row plan: entry z = blockIdx.y is one block pair of a job of its own (source, scores, summary), work slot z.
// Every run of raw scores is written (no ballots, no thresholds).
constexpr int kMaxTailBatch = 8;
struct TailBatch {
    int n;
    const void* src[kMaxTailBatch];      // the haystack's samples from the tail's first score on
    long long src_len[kMaxTailBatch];
    float* dst[kMaxTailBatch];           // out_count scores
    long long out_count[kMaxTailBatch];
    float2* stats32[kMaxTailBatch];      // their level-0 summary
};
// hop, src_kind: shared by the entries (one needle, one sample format)
hipError_t launch_tail_batch_k1(hipStream_t st, const TailBatch& tb, int hop, int src_kind, float2* work, const PlanDev& pl, int half);
hipError_t launch_tail_batch_k3(hipStream_t st, const TailBatch& tb, int hop, const float2* work, const PlanDev& pl, float out_scale, int half);
// A tail's scores and summary into the score-side buffers the pick reads, and the main layout's ballots / thresholds of
// the block they belong to set to "every run written" (one small launch on the pick's stream)
// several needles (K3Group): the ballots and thresholds of block `blk` of the main layout set to "every run written"
// for every needle of the group, one launch
hipError_t launch_tail_preset_group(hipStream_t st, const K3Group& grp, long long blk, int log_n1, int log_n2);
hipError_t launch_tail_commit(hipStream_t st, const float* tail_scores, float* scores, long long n, const float2* tail_stats32, float2* stats32,
                              unsigned long long* wbits, long long words, float* theta, int tiles);
hipError_t launch_k3_group(hipStream_t st, const Job& job, int npairs, const K3Group& grp, const PlanDev& pl, const ScanCfg& scan);
// needles of at most this many samples are correlated by direct summation (no transform)
constexpr int kDirectMaxNeedle = 64;
hipError_t launch_direct(hipStream_t st, const Job& job, const float* needle, int s, float out_scale);
bool plan_is_r16(const PlanDev& pl);
bool plan_is_c512(const PlanDev& pl);    // N = 2^22 = 512 x 8192, 512-thread column kernels
bool plan_is_c512w(const PlanDev& pl);   // N = 2^23 = 512 x 16384: the 512-row column kernels on longer rows (measurement only: no row kernel yet)
bool plan_is_c1024(const PlanDev& pl);   // N = 2^23 = 1024 x 8192, 1024-thread column kernels (f32 work matrix only)
bool plan_has_scan(const PlanDev& pl);   // K3 of this plan carries the fused score scan
bool plan_k2_is_r16(const PlanDev& pl);
hipError_t fft_kernels_init();

// flags[r] = 1 if x[ranges[r].a .. ranges[r].b) holds a non-finite sample (f32 sources only)
hipError_t launch_nonfinite_ranges(hipStream_t st, const float* x, const Segment* ranges, int nranges, int* flags);
// bad (optional, host-visible word): set to 1 when a score is not finite
hipError_t launch_tile_stats(hipStream_t st, const float* g, long long n, float2* stats, int* bad = nullptr);
// The picks of a needle group as one set of launches (several needles against one haystack): needle z (blockIdx.y / .z)
// works on its own score array, summaries, flags and thresholds; chunk list and geometry are shared, results are laid
// out needle after needle (n = 0: an ordinary, single pick).
struct PickGroup {
    int n;
    const float* g[kMaxNeedleGroup];
    float2* stats[kMaxNeedleGroup];
    const float2* stats32[kMaxNeedleGroup];
    const unsigned long long* wbits[kMaxNeedleGroup];
    const float* theta[kMaxNeedleGroup];
    int hdr_off[kMaxNeedleGroup];   // result headers of needle z: hdr + hdr_off[z] (relative to the pointer the launch is given)
};
hipError_t launch_stats_reduce(hipStream_t st, const float2* stats32, long long n, float2* stats, int* bad = nullptr,
                               const PickGroup* grp = nullptr);
// per-chunk result header: the count, an overflow flag and the first few peaks
// inline, so that the common case needs a single small device-to-host copy
constexpr int kInlinePeaks = 4;
struct SegHeader {
    int n;
    int overflow;      // bit 0: more than AM_MAX_PEAKS_PER_CHUNK peaks; bit 1: a write threshold was too high for this chunk;
                       // bit 2: more than kInlinePeaks peaks and no room left in the spill arena
    float seg_min;     // (lower bound of the) chunk minimum, for adapting theta
    int arena_off;     // n > kInlinePeaks: the whole list sits at arena.base[arena_off .. arena_off + n)
    am_peak first[kInlinePeaks];
};
// Spill area for the peak lists of chunks with more than kInlinePeaks peaks: a bump
// allocator shared by every chunk of a call (base may be host-mapped memory; cursor is
// device memory zeroed before the first pick of the call).  base == nullptr: no spill.
struct PeakArena {
    am_peak* base;
    unsigned* cursor;
    unsigned cap;
};
// Which raw scores exist (K3 writes them sparsely, run by run): wbits == nullptr means all.
struct SparseScores {
    const unsigned long long* wbits;   // see ScanCfg
    const float2* stats32;
    const float* tile_theta;           // thresholds K3 used (the pick's certificate reads them)
    int hop, log_n2, log_n1;           // block geometry: score n of a block = row (n >> log_n2), column (n & (2^log_n2 - 1))
    double inv_hop;
    int* redo_pairs;                   // per block pair: set by the pick when a chunk fed by the pair fails its certificate (or null)
    unsigned char* fail_flags;         // host-visible, one per chunk of the launch: set when the chunk failed its certificate (or null)
};
// Hand-over of chunks with many candidate tiles from peaks_kernel to peaks_wide /
// peaks_finish (device memory, one entry per chunk of the launch; list: AM_MAX_PEAKS_PER_CHUNK
// entries per chunk).  list == nullptr: every chunk is finished by its one workgroup.
struct WideState {
    int* state;          // 0 = finished by peaks_kernel; bit 0 = handed over, bits 1 / 2 = scan the raw head / tail piece
    unsigned* count;     // peaks appended to the chunk's list (> AM_MAX_PEAKS_PER_CHUNK: overflow)
    float* seg_min;
    unsigned long long* best;   // min_distance >= chunk length: running maximum of the peaks that pass (order-preserving key)
    am_peak* list;
    unsigned cap;        // entries per chunk in list (AM_MAX_PEAKS_PER_CHUNK on the usual path)
    int* ntiles;         // candidate tiles listed for the chunk (-1: not listed, the parts test every tile)
    int* tiles;          // kWideTileList entries per chunk: candidate tiles relative to the chunk's first full tile
};
constexpr int kWideTileList = 1024;
// The rules of find_peaks 0.1 that nothing available offline pins (crate source absent, no reference test; SURVEY.md
// 8c): the defaults are the library's documented choice (oracle/oracle.c), every alternative is implemented in the
// kernels AND in the checker, so that whoever has the crate flips an option instead of rewriting a kernel.
struct PeakPolicy {
    int order;        // 0: prominence filter, then distance filter (default); 1: distance first, then prominence (scipy's order)
    int inclusive;    // the distance filter drops a peak whose distance to a kept, higher one is  0: < min_distance (default)  1: <= min_distance
    int from_start;   // ... measured between  0: plateau middles (start + end) / 2 (default)  1: plateau starts
};
// A chunk with more than AM_MAX_PEAKS_PER_CHUNK peaks passing the prominence filter (rare: a
// min_distance shorter than the chunk and a tiny prominence bound).  launch_peaks_wide_one runs the
// list-building kernel for ONE chunk (wide.state[0] must be 1; wide.cap = 0 only counts);
// launch_peaks_big_finish orders that list by (height descending, position ascending) with a
// radix sort in global memory and applies the greedy min_distance filter through a table of
// min_distance-wide buckets (each holds at most one kept peak).  Scratch: keys 2 x n x 8 bytes,
// idx 2 x n x 4 bytes, table ((b - a) / min_distance + 3) x 8 bytes preset to 0xFF.
hipError_t launch_peaks_wide_one(hipStream_t st, const float* g, long long g_len, const float2* stats, const Segment* d_seg,
                                 float min_prom, long long min_dist, const SparseScores& sp, const WideState& wide, const PeakPolicy& pol);
hipError_t launch_peaks_big_finish(hipStream_t st, const am_peak* list, unsigned n, long long a, long long min_dist,
                                   unsigned long long* keys, unsigned* idx, long long* table, am_peak* out, unsigned* out_n,
                                   const PeakPolicy& pol);
// only_failed: pick only the chunks whose header says "certificate failed" (after the device-side redo of K3)
hipError_t launch_peaks(hipStream_t st, const float* g, long long g_len, const float2* stats,
                        const Segment* d_segs, int nsegs, float min_prom, long long min_dist,
                        am_peak* d_out, SegHeader* d_hdr, const SparseScores& sp, const PeakArena& arena,
                        const WideState& wide, bool only_failed, const PeakPolicy& pol, const PickGroup* grp = nullptr);
// writes sumsq_parts(n) partial sums (one per workgroup) to d_parts
int sumsq_parts(long long n);
hipError_t launch_sumsq(hipStream_t st, const float* x, long long n, double* d_parts);
hipError_t launch_synth(hipStream_t st, float* out, uint32_t seed, uint32_t stream, uint64_t first,
                        long long n, float amp);
hipError_t launch_axpy(hipStream_t st, float* dst, const float* src, long long n, float gain);
hipError_t launch_synth_pcm16(hipStream_t st, int16_t* out, uint32_t seed, uint32_t stream, uint64_t first, long long frames, float amp);
hipError_t launch_add_pcm16(hipStream_t st, int16_t* dst, const int16_t* src, long long frames);
hipError_t launch_pcm_downmix(hipStream_t st, const int16_t* in, long long frames, float* out);

}  // namespace am
