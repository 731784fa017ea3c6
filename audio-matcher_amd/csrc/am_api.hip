// am_api.hip -- host side of libaudiomatch_amd.so: device contexts, transform
// plans, the needle handle, the overlap-save engine, the chunk driver and the C
// ABI of include/audiomatch.h.
//
// Host-side mirror of the reference's driver (paths relative to the reference):
//   calc_chunks            src/matcher/audio_matcher.rs:88-141
//   is_overshadowed        src/matcher/audio_matcher.rs:143-160
//   start_as_duration      src/matcher/mod.rs:127-129
//   Mode crop / centered   src/matcher/audio_matcher.rs:450-464
// All arithmetic on samples runs in the HIP kernels of am_fft.hip /
// am_peaks.hip; there is no CPU fallback.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <condition_variable>
#include <vector>

#include "am_kernels.h"

namespace am {

// ---------------------------------------------------------------------------
static thread_local std::string t_err;

static int fail(int code, const std::string& msg) {
    t_err = msg;
    return code;
}
static int hip_fail(hipError_t e, const char* what) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    t_err = buf;
    return e == hipErrorOutOfMemory ? AM_ERR_OOM : AM_ERR_HIP;
}
#define AM_HIP(call)                                         \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #call);    \
    } while (0)

// progress hooks (audio_matcher.rs:102-117, 129); a call works on the snapshot it takes on entry
struct Hooks {
    am_progress_fn fn = nullptr;
    void* user = nullptr;
    am_chunk_progress_fn chunk_fn = nullptr;
    void* chunk_user = nullptr;
};
static std::mutex g_hooks_mu;
static Hooks g_hooks;
static Hooks snapshot_hooks() {
    std::lock_guard<std::mutex> lk(g_hooks_mu);
    return g_hooks;
}

// Process-wide option DEFAULTS (am_set_option).  Every entry point reads them once, on
// entry, into an Opts value that the whole call then works with, so a concurrent
// am_set_option never changes a call half way; "log_n" and "half_pipeline" can also be
// fixed per needle handle (am_needle_set_option), which wins over the default.
static std::atomic<long long> g_opt_log_n{0};            // 0 = auto
static std::atomic<long long> g_opt_pairs_per_group{64};
static std::atomic<long long> g_opt_profile_mask{-1};    // bit i = bracket kernel class i with events while profiling is on
static std::atomic<long long> g_opt_profile_every{1};    // ... every n-th launch of the class only (an event pair costs the stream about 8 us per kernel boundary)
static std::atomic<long long> g_opt_half{0};             // 1 = half-precision storage of the work matrix (config 5)
static std::atomic<long long> g_opt_batch_overlap{1};    // 1 = in a batch, pick the peaks of haystack k beside the transforms of k+1
static std::atomic<long long> g_opt_needle_group{8};     // needles sharing one forward row transform in am_match_multi_device
static std::atomic<long long> g_opt_pick_priority{0};    // 1 = the pick's stream is created with the lowest priority (read at context creation)
static std::atomic<long long> g_opt_pick_group{1};       // 1 = ... and so do the group's picks (0: four small launches per needle, for A/B)
static std::atomic<long long> g_opt_k3_group{1};         // 1 = the K3s of a needle group run as one launch (0: one launch per needle, for A/B)
static std::atomic<long long> g_opt_host_pick_wait{1};    // 1 = a batch's host thread waits for the pick that last read a score set before it queues the next haystack into it (0: the stream waits)
static std::atomic<long long> g_opt_device_redo{1};      // 0 = failed certificates are redone by the host path only (experiments)
static std::atomic<long long> g_opt_tail_block{1};       // 1 = a haystack's last, odd block goes through the next smaller plan (TailPlan); 0 = as half of a full pair
static std::atomic<long long> g_opt_dense{0};            // 1 = K3 writes every raw score (theta = -inf): the worst case of the sparse-score path
// test hooks (defaults = production behaviour)
static std::atomic<long long> g_opt_debug_no_realloc{0};     // 1 = a scratch buffer that would be (re)allocated while a call is queueing fails the call
static std::atomic<long long> g_opt_debug_redo_arm_at{-2};   // >= 0: the device-side redo of a batch arms at that haystack; -1: never; -2: when a failure is seen
// The semantics nothing available offline pins (the crates find_peaks 0.1 and common are absent, no
// reference test covers these rules).  Defaults = the documented choices of oracle/oracle.c; every alternative exists
// in the kernels, on the host AND in the checker, so that one run by someone who has the crates settles each with an
// option instead of a rewrite (DESIGN.md section 3 lists inputs on which the variants differ).
static std::atomic<long long> g_opt_peak_filter_order{0};   // 0 = prominence, then distance; 1 = distance, then prominence (scipy's order)
static std::atomic<long long> g_opt_distance_rule{0};       // bit 0: drop at distance <= min_distance (default <); bit 1: between plateau starts (default middles)
static std::atomic<long long> g_opt_tail_window{0};         // 0 = chunked() emits the shorter windows at the end; 1 = only full-length windows
static std::atomic<long long> g_opt_surrounding_from{0};    // filter_surrounding's neighbours: 0 = of the sorted, unfiltered sequence; 1 = the neighbour before is the last element kept
struct Opts {
    long long log_n, pairs_per_group, half, batch_overlap, needle_group, dense, device_redo, debug_no_realloc, debug_redo_arm_at;
    long long peak_filter_order, distance_rule, tail_window, surrounding_from, k3_group, pick_group, tail_block, host_pick_wait;
    PeakPolicy peak_policy() const { return PeakPolicy{(int)peak_filter_order, (int)(distance_rule & 1), (int)((distance_rule >> 1) & 1)}; }
};
static const float kHalfGain = 1024.0f;      // keeps the stored values of a normalised score near 1
static const double kMinEfficiency = 0.75;  // hop / N the auto plan accepts
static const int kLogNMin = 10, kLogNMax = 23;
// needles longer than this run on N = 2^22 (measured crossover between 2 and 5 s of 44.1 kHz
// audio, tools/needle_sweep.py, profiles/r03/needle_sweep.txt: 2 s 0.674 against 0.685 ms per hour of
// audio, 5 s 0.725 against 0.702)
static const long long kWideFromSamples = 140000;
// needles longer than this run on N = 2^23 = 1024 x 8192 (measured crossover between 30 and 36 s of 44.1 kHz
// audio, profiles/r03/needle_sweep.txt: the 1024-row column kernels cost more per point, the hop is longer)
static const long long kWidestFromSamples = 1500000;
// Needles longer than this (half a 2^23 transform) are cut into segments of at most 2^22 samples:
// corr(hay, needle)[j] = sum_i corr(hay, segment_i)[j + offset_i], every segment on the register kernels
// of the 2^23 plan (hop efficiency of at least one half), the partial sums added up in the score array by
// K3 (MyConvolve::correlate accepts any length, audio_matcher.rs:414-457).
static const long long kSegmentFrom = 1ll << 22;
static const long long kSegmentLen = 1ll << 22;

// ---------------------------------------------------------------------------
// While a batch is being queued (kernels of earlier haystacks still running, or not yet started) no
// scratch buffer may move: hipFree waits for the device (the overlap of pick and transforms stalls) and a
// buffer whose contents a later launch still expects would be lost (round 3, gpurun_out/r03q: a peak lost
// to a flag buffer re-allocated under a running pick).  match_many / match_multi_many size everything
// before their queueing loops; with the option "debug_no_realloc" an ensure() that would still have to
// allocate inside such a loop fails the call instead (tests/test_gpu_round4.py).
static thread_local int t_no_realloc = 0;
struct QueueingScope {
    bool on;
    explicit QueueingScope(bool enable) : on(enable) { if (on) ++t_no_realloc; }
    ~QueueingScope() { end(); }
    void end() { if (on) { --t_no_realloc; on = false; } }
    QueueingScope(const QueueingScope&) = delete;
    QueueingScope& operator=(const QueueingScope&) = delete;
};
static int realloc_refused(const char* what, size_t bytes, size_t cap) {
    char buf[160];
    snprintf(buf, sizeof(buf), "debug_no_realloc: %s buffer would grow from %zu to %zu bytes while a call is queueing", what, cap, bytes);
    return fail(AM_ERR_HIP, buf);
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return AM_OK;
        if (t_no_realloc > 0) return realloc_refused("a device", bytes, cap);
        release();
        size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
            if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc(scratch)"); }
        }
        cap = want;
        return AM_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    unsigned flags = hipHostMallocDefault;
    int ensure(size_t bytes) {
        if (bytes <= cap) return AM_OK;
        if (t_no_realloc > 0) return realloc_refused("a pinned host", bytes, cap);
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes, flags);
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipHostMalloc"); }
        cap = bytes;
        return AM_OK;
    }
};

struct Plan {
    PlanDev dev{};
    float2* tables = nullptr;  // one allocation holding the four tables
    unsigned* mf = nullptr;    // constant tables of the matrix-core row kernel (N2 = 8192 only)
};

struct ProfRec { int name; hipEvent_t e0, e1; };
static const char* kKernelNames[] = {"k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks", "other"};
enum { KN_K1 = 0, KN_K2, KN_K3, KN_STATS, KN_PEAKS, KN_OTHER, KN_COUNT };

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // peak pick of haystack k beside the transforms of k+1 (batches)
    hipStream_t stream_tail = nullptr;       // a haystack's odd last block on the smaller plan, beside its main pass (run_tail_block)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf work_tail, tail_scores, tail_stats;   // (a batch computes the tails of up to kMaxTailBatch haystacks per launch: two alternating halves)
    DevBuf work_tail2;                           // several needles: the tail's inverse rows, one matrix per needle of a group
    std::recursive_mutex mu;
    std::map<int, Plan> plans;
    DevBuf work, work2, scores, stats, stats32, wflags, segs, peaks, io_in, io_out, sum, arena_cur, wide_ctl, wide_list, wide_tiles;
    // second set of the score-side buffers: in a batch the peak pick of haystack k runs on
    // stream2 beside the transforms of haystack k+1, which then need their own set
    DevBuf scores_b, stats_b, stats32_b, wflags_b, peaks_b;
    // device-side redo (batches): a second work matrix, so that the inverse rows of haystack k are still there
    // when its pick has found chunks whose certificate failed, and the per-pair "run again" flags of both sets
    DevBuf work_b, redo_pairs[2];
    // several needles: the K3s of a needle group run as ONE launch, every needle of the group with score-side
    // buffers of its own; two such sets alternate (the picks of group g beside the transforms of group g + 1)
    DevBuf grp_scores[2 * kMaxNeedleGroup], grp_stats32[2 * kMaxNeedleGroup], grp_wflags[2 * kMaxNeedleGroup];
    DevBuf grp_stats[kMaxNeedleGroup];   // tile summaries of the group's picks (one set: picks run one group after the other)
    HostBuf failcnt;   // host-visible: one byte per chunk of a call, set when the chunk failed its certificate
    hipEvent_t ev_k3[2] = {nullptr, nullptr}, ev_pick[2] = {nullptr, nullptr};
    HostBuf pinned;
    // Per-chunk result headers live in coherent pinned host memory that the peak
    // kernel writes directly (a few KB per haystack): no device-to-host copy
    // sits between the last kernel and the host's wake-up.
    HostBuf hdr;
    HostBuf spill;   // spill arena of the single-chunk passes (same kind of memory)
    HostBuf badflag; // one word per haystack of a call: "some score was not finite"
    DevBuf ranges, range_flags;   // work area of the non-finite-sample search (rare path)
    DevBuf big;                   // lists, sort keys and bucket table of a chunk with more than AM_MAX_PEAKS_PER_CHUNK peaks (rare path)
    // the chunk list currently resident in `segs` (re-uploaded only when it changes)
    std::vector<Segment> segs_resident;
    // profiling
    bool prof = false;
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    double prof_ms[KN_COUNT] = {0};
    uint64_t prof_n[KN_COUNT] = {0};
    uint64_t prof_seq[KN_COUNT] = {0};   // launches of the class seen while profiling is on (option profile_every)
};

// Events that order the library's streams of ONE device among themselves (the pick behind K3, K3 behind the pick that
// last read its score set, the tail stream) and the events that time kernels: without the system-scope fence a default
// event performs when it is recorded (a write-back and invalidation of the caches).  Nothing here needs that fence:
// kernel boundaries order device memory by themselves, and what the host reads (result headers in pinned memory) it
// reads behind a hipStreamSynchronize.  Measured -0.7 % on the headline, near the noise: what an event costs a stream
// is its barrier packet, 4 - 8 us of a kernel boundary, fence or not (profiles/r04/event_gaps.txt).
#ifndef AM_EVENT_NO_SYSTEM_FENCE
#define AM_EVENT_NO_SYSTEM_FENCE 1
#endif
static const unsigned kSyncEvent = hipEventDisableTiming | (AM_EVENT_NO_SYSTEM_FENCE ? hipEventDisableSystemFence : 0u);
static std::mutex g_ctx_mu;
static std::map<int, Ctx*> g_ctx;

static int get_ctx(int device, Ctx** out) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(AM_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= n) return fail(AM_ERR_NO_DEVICE, "device ordinal out of range");
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) { *out = it->second; AM_HIP(hipSetDevice(device)); return AM_OK; }
    AM_HIP(hipSetDevice(device));
    (void)hipSetDeviceFlags(hipDeviceScheduleSpin);   // may fail if the primary context is already active: harmless
    (void)hipGetLastError();
    AM_HIP(fft_kernels_init());   // function attributes are per device
    Ctx* c = new Ctx();
    c->device = device;
    c->hdr.flags = hipHostMallocMapped | hipHostMallocCoherent;
    c->failcnt.flags = hipHostMallocMapped | hipHostMallocCoherent;
    c->spill.flags = hipHostMallocMapped | hipHostMallocCoherent;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return hip_fail(se, "hipStreamCreate"); }
    // the pick's stream: small, latency-bound kernels that run beside the next haystack's transforms; at the lowest
    // priority their workgroups fill what the transform kernels leave free instead of competing for dispatch slots
    // (option "pick_stream_priority", read when the context is created: 0 = same priority as the transforms)
    {
        int least = 0, greatest = 0;
        if (g_opt_pick_priority.load(std::memory_order_relaxed) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            (void)hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, least);
        else
            (void)hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
        (void)hipGetLastError();
    }
    (void)hipStreamCreateWithFlags(&c->stream_tail, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&c->ev_fork, kSyncEvent);
    (void)hipEventCreateWithFlags(&c->ev_join, kSyncEvent);
    (void)hipGetLastError();
    for (int i = 0; i < 2; ++i) {
        (void)hipEventCreateWithFlags(&c->ev_k3[i], kSyncEvent);
        (void)hipEventCreateWithFlags(&c->ev_pick[i], kSyncEvent);
    }
    g_ctx[device] = c;
    *out = c;
    return AM_OK;
}

// ---- profiling helpers ------------------------------------------------------
static hipEvent_t prof_event(Ctx* c) {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, AM_EVENT_NO_SYSTEM_FENCE ? hipEventDisableSystemFence : hipEventDefault);   // (timing only: see kSyncEvent)
    return e;
}
struct ProfScope {
    Ctx* c; int name; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on;
    ProfScope(Ctx* c_, int name_, hipStream_t st_ = nullptr) : c(c_), name(name_), st(st_ ? st_ : c_->stream) {
        on = c->prof && ((g_opt_profile_mask.load(std::memory_order_relaxed) >> name) & 1);
        if (on) {
            const long long every = std::max<long long>(1, g_opt_profile_every.load(std::memory_order_relaxed));
            on = (c->prof_seq[name]++ % (uint64_t)every) == 0;
        }
        if (on) { e0 = prof_event(c); e1 = prof_event(c); (void)hipEventRecord(e0, st); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(e1, st); c->pending.push_back({name, e0, e1}); }
    }
};
static void prof_harvest(Ctx* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream_tail) (void)hipStreamSynchronize(c->stream_tail);
    for (auto& r : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { c->prof_ms[r.name] += ms; c->prof_n[r.name] += 1; }
        c->pool.push_back(r.e0); c->pool.push_back(r.e1);
    }
    c->pending.clear();
}

// ---- copies ------------------------------------------------------------------
// Every copy of the library runs on the context's stream and is waited for there.
// That stream is non-blocking, i.e. not ordered with the null stream a plain
// hipMemcpy uses; a device-to-device hipMemcpy returns before the copy has run and
// a copy from pageable host memory may return once the data is staged, so kernels
// queued on the context's stream right afterwards could otherwise read data that has
// not arrived yet.
static hipError_t copy_on_stream(Ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(c->stream);
}

// ---- plans --------------------------------------------------------------------
static void fill_twiddles(std::vector<float2>& v, size_t off, size_t count, double denom, double mult) {
    for (size_t k = 0; k < count; ++k) {
        const double ang = -2.0 * M_PI * (double)k * mult / denom;
        v[off + k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
}

// Constant tables of k2_rows_m16 (am_fft.hip): the DFT-16 and DFT-32 matrices as operands of
// v_mfma_f32_16x16x32_f16 -- lane (i = lane & 15, g = lane >> 4) holds row i, k = 8g .. 8g+7 with k = 2 p' + {re, im}
// of input point p = 4g + p' (+ 16 ks): [Re F | -Im F] rows give the outputs' real parts, [Im F | Re F] the imaginary
// parts, F[m][p] = W^(m p) -- and every thread's twiddles as h2: T1[gl][e][r] = W_8192^((32 (4w + gl) + 2n + e)(4g + r)),
// T2[ch][r] = W_512^((16 ch + n)(4g + r)) for thread t = 64 w + 16 g + n.  Values are computed in f64 and rounded once.
static void build_mfma_tables(std::vector<unsigned>& tab) {
    tab.assign((size_t)k2_mfma_table_dwords(), 0u);
    auto pack = [](double re, double im) {
        const _Float16 a = (_Float16)re, b = (_Float16)im;
        unsigned short ua, ub;
        memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
        return (unsigned)ua | ((unsigned)ub << 16);
    };
    // operand element pair (k = 2p', 2p'+1) of row m for input point p: real-part rows (cos, sin), imaginary-part rows (-sin, cos)
    // with F = cos - i sin:  re_out = sum cos x_re + sin x_im,  im_out = sum -sin x_re + cos x_im
    auto operand = [&](size_t base, int m_off, int p_off, double denom) {
        for (int ri = 0; ri < 2; ++ri)
            for (int lane = 0; lane < 64; ++lane)
                for (int pp = 0; pp < 4; ++pp) {
                    const int m = m_off + (lane & 15), pt = p_off + 4 * (lane >> 4) + pp;
                    const double ang = 2.0 * M_PI * (double)((m * pt) % (int)denom) / denom;
                    tab[base + (size_t)ri * 256 + (size_t)lane * 4 + pp] = ri == 0 ? pack(std::cos(ang), std::sin(ang)) : pack(-std::sin(ang), std::cos(ang));
                }
    };
    operand(0, 0, 0, 16.0);                                                   // A16: re rows, im rows
    for (int mb = 0; mb < 2; ++mb)
        for (int ks = 0; ks < 2; ++ks) operand(512 + (size_t)(mb * 2 + ks) * 512, 16 * mb, 16 * ks, 32.0);   // A32[mb][ks][re, im]
    const size_t t1 = 512 + 2048, t2 = t1 + 256 * 32;
    for (int t = 0; t < 256; ++t) {
        const int w = t >> 6, g = (t >> 4) & 3, n = t & 15;
        for (int gl = 0; gl < 4; ++gl)
            for (int e = 0; e < 2; ++e)
                for (int r = 0; r < 4; ++r) {
                    const long long m = ((long long)(32 * (4 * w + gl) + 2 * n + e) * (4 * g + r)) % 8192;
                    const double ang = -2.0 * M_PI * (double)m / 8192.0;
                    tab[t1 + (size_t)t * 32 + gl * 8 + e * 4 + r] = pack(std::cos(ang), std::sin(ang));
                }
        for (int ch = 0; ch < 2; ++ch)
            for (int r = 0; r < 4; ++r) {
                const int m = ((16 * ch + n) * (4 * g + r)) % 512;
                const double ang = -2.0 * M_PI * (double)m / 512.0;
                tab[t2 + (size_t)t * 8 + ch * 4 + r] = pack(std::cos(ang), std::sin(ang));
            }
    }
}

// force_logN1: another factorisation than the production one (am_debug_column_bench: 2^23 as 512 x 16384)
static int get_plan(Ctx* c, int logN, const Plan** out, int force_logN1 = 0) {
    const int key = force_logN1 ? 1000 * force_logN1 + logN : logN;
    auto it = c->plans.find(key);
    if (it != c->plans.end()) { *out = &it->second; return AM_OK; }
    if (logN < kLogNMin || logN > kLogNMax) return fail(AM_ERR_INVALID_ARG, "unsupported transform size");
    Plan p;
    int logN1 = logN - 13;
    if (logN1 < kColsLog) logN1 = kColsLog;
    if (logN1 > 10) logN1 = 10;
    if (force_logN1) logN1 = force_logN1;
    int logN2 = logN - logN1;
    // N = 2^21 -> 256 x 8192, N = 2^22 -> 512 x 8192, N = 2^23 -> 1024 x 8192: the register kernels
    const int logLo = (logN + 1) / 2;
    const size_t n1h = (size_t)1 << (logN1 - 1), n2h = (size_t)1 << (logN2 - 1);
    const size_t nlo = (size_t)1 << logLo, nhi = (size_t)1 << (logN - logLo);
    // float2 tables, then the float4 ones (see PlanDev): offsets in float2 units, the float4 part 16-byte aligned
    const size_t f2count = (n1h + n2h + nlo + nhi + 1) & ~(size_t)1;
    const size_t nk2j = logN2 == 13 ? 2 * 256 : 0, nk2c = logN2 == 13 ? 2 * 16 : 0;
    std::vector<float2> host(f2count + 2 * (nlo + nhi + nk2j + nk2c));
    fill_twiddles(host, 0, n1h, (double)(1u << logN1), 1.0);
    fill_twiddles(host, n1h, n2h, (double)(1u << logN2), 1.0);
    fill_twiddles(host, n1h + n2h, nlo, (double)((size_t)1 << logN), 1.0);
    fill_twiddles(host, n1h + n2h + nlo, nhi, (double)((size_t)1 << logN), (double)nlo);
    auto tw = [](double num, double denom) {
        const double ang = -2.0 * M_PI * std::fmod(num, denom) / denom;
        return make_float2((float)std::cos(ang), (float)std::sin(ang));
    };
    const double dN = (double)((size_t)1 << logN);
    size_t o = f2count;
    const size_t o_lo4 = o;
    for (size_t k = 0; k < nlo; ++k) { host[o++] = tw((double)k, dN); host[o++] = tw(4.0 * (double)k, dN); }
    const size_t o_hi4 = o;
    for (size_t k = 0; k < nhi; ++k) { host[o++] = tw((double)k * (double)nlo, dN); host[o++] = tw(4.0 * (double)k * (double)nlo, dN); }
    const size_t o_k2j = o;
    for (size_t t = 0; t < nk2j / 2; ++t) {
        host[o++] = tw(2.0 * t, 8192.0); host[o++] = tw(2.0 * t + 1.0, 8192.0);
        host[o++] = tw(8.0 * t, 8192.0); host[o++] = tw(8.0 * t + 4.0, 8192.0);
    }
    const size_t o_k2c = o;
    for (size_t cidx = 0; cidx < nk2c / 2; ++cidx) {
        host[o++] = tw(32.0 * cidx, 8192.0); host[o++] = tw(32.0 * cidx + 16.0, 8192.0);
        host[o++] = tw(128.0 * cidx, 8192.0); host[o++] = tw(128.0 * cidx + 64.0, 8192.0);
    }
    AM_HIP(hipMalloc((void**)&p.tables, host.size() * sizeof(float2)));
    AM_HIP(copy_on_stream(c, p.tables, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice));
    p.dev.logN = logN; p.dev.logN1 = logN1; p.dev.logN2 = logN2; p.dev.logLo = logLo;
    p.dev.tw1 = p.tables;
    p.dev.tw2 = p.tables + n1h;
    p.dev.twlo = p.tables + n1h + n2h;
    p.dev.twhi = p.tables + n1h + n2h + nlo;
    p.dev.twlo4 = reinterpret_cast<const float4*>(p.tables + o_lo4);
    p.dev.twhi4 = reinterpret_cast<const float4*>(p.tables + o_hi4);
    p.dev.k2j = nk2j ? reinterpret_cast<const float4*>(p.tables + o_k2j) : nullptr;
    p.dev.k2c = nk2c ? reinterpret_cast<const float4*>(p.tables + o_k2c) : nullptr;
    p.dev.mf = nullptr;
    if (logN2 == 13) {
        std::vector<unsigned> tab;
        build_mfma_tables(tab);
        AM_HIP(hipMalloc((void**)&p.mf, tab.size() * sizeof(unsigned)));
        AM_HIP(copy_on_stream(c, p.mf, tab.data(), tab.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        p.dev.mf = p.mf;
    }
    auto ins = c->plans.emplace(key, p);
    *out = &ins.first->second;
    return AM_OK;
}

}  // namespace am

// ---------------------------------------------------------------------------
struct am_needle {
    am::Ctx* ctx = nullptr;
    float* d_needle = nullptr;
    size_t n = 0;
    float inv_autocorr = 0.f;
    std::map<int, float2*> spectra;  // logN -> conj(H)/N in pipeline layout
    std::map<int, unsigned*> spectra16;   // logN -> the same as scaled __half2 points (half_pipeline = 2)
    std::map<int, unsigned*> spectra16m;  // logN -> the same conjugated, in [a'][b'][c'] order (option k2_mfma)
    // Lowest chunk minimum of each of the last few haystacks matched with this needle (index 0:
    // unscaled scores, 1: AM_SCALE_LIB).  Bounds the raw-score write threshold from above, so that a
    // score array that drifts slowly (chunk minimum in another block pair than a tile's scores)
    // stays inside its certificate; a ring, so that one unusual haystack is forgotten again.
    static constexpr int kRecent = 8;
    float recent_min[2][kRecent];
    int recent_n[2] = {0, 0}, recent_pos[2] = {0, 0};
    void remember_min(int sm, float v) {
        recent_min[sm][recent_pos[sm]] = v;
        recent_pos[sm] = (recent_pos[sm] + 1) % kRecent;
        if (recent_n[sm] < kRecent) ++recent_n[sm];
    }
    // haystacks left for which the ring takes the LOWEST chunk minimum (after a haystack in which many chunks
    // failed their certificate: a drifting score array); otherwise, where a failed chunk is redone on the
    // device, it takes the median -- the background level -- so that a few chunks with deep dips (a hit whose
    // autocorrelation has negative lobes) do not make every later haystack write all its scores
    int conservative_left[2] = {0, 0};
    // haystacks left for which a batch queues the device-side redo (a K3 launch that looks at the pairs' flags and
    // a second pick per haystack: 1.6 % of the headline's time when nothing ever fails).  Armed by a failed
    // certificate -- of an earlier call, or of an earlier haystack of the same call as soon as its flag has
    // arrived in host memory; until then such a chunk is redone from the host, as in single calls.
    int redo_armed_left[2] = {0, 0};
    float hist_min(int sm) const {
        float m = FLT_MAX;
        for (int i = 0; i < recent_n[sm]; ++i) m = std::min(m, recent_min[sm][i]);
        return m;
    }
    // per-handle overrides of the process-wide option defaults (-1 = follow the default)
    long long opt_log_n = -1, opt_half = -1;
    // Needle partitioning (needles longer than kSegmentFrom samples): sub-handles over slices of d_needle
    // (not owned), each with its own spectra; segment i starts at sample seg_off[i] of the needle.
    std::vector<am_needle*> segments;
    std::vector<long long> seg_off;
    bool owns_data = true;
};

namespace am {

static Opts snapshot_opts(const am_needle* h) {
    Opts o;
    o.log_n = (h && h->opt_log_n >= 0) ? h->opt_log_n : g_opt_log_n.load(std::memory_order_relaxed);
    o.half = (h && h->opt_half >= 0) ? h->opt_half : g_opt_half.load(std::memory_order_relaxed);
    o.pairs_per_group = g_opt_pairs_per_group.load(std::memory_order_relaxed);
    o.batch_overlap = g_opt_batch_overlap.load(std::memory_order_relaxed);
    o.needle_group = g_opt_needle_group.load(std::memory_order_relaxed);
    o.dense = g_opt_dense.load(std::memory_order_relaxed);
    o.device_redo = g_opt_device_redo.load(std::memory_order_relaxed);
    o.debug_no_realloc = g_opt_debug_no_realloc.load(std::memory_order_relaxed);
    o.debug_redo_arm_at = g_opt_debug_redo_arm_at.load(std::memory_order_relaxed);
    o.peak_filter_order = g_opt_peak_filter_order.load(std::memory_order_relaxed);
    o.distance_rule = g_opt_distance_rule.load(std::memory_order_relaxed);
    o.tail_window = g_opt_tail_window.load(std::memory_order_relaxed);
    o.surrounding_from = g_opt_surrounding_from.load(std::memory_order_relaxed);
    o.k3_group = g_opt_k3_group.load(std::memory_order_relaxed);
    o.pick_group = g_opt_pick_group.load(std::memory_order_relaxed);
    o.tail_block = g_opt_tail_block.load(std::memory_order_relaxed);
    o.host_pick_wait = g_opt_host_pick_wait.load(std::memory_order_relaxed);
    return o;
}

static int pick_log_n(size_t s, long long out_count, const Opts& o, int* logN_out) {
    // smallest transform that can hold the needle at all
    int min_log = kLogNMin;
    while (min_log <= kLogNMax && ((size_t)1 << min_log) < s + 1) ++min_log;
    if (min_log > kLogNMax) return fail(AM_ERR_INVALID_ARG, "needle too long for a forced transform size (2^23 at most; leave log_n at 0 for needle partitioning)");
    if (o.log_n > 0) {
        int l = (int)o.log_n;
        if (l < min_log) l = min_log;
        if (l > kLogNMax) l = kLogNMax;
        *logN_out = l;
        return AM_OK;
    }
    const long long span = out_count + (long long)s - 1;
    // The register-resident kernels exist for N = 2^21 and 2^22 only and are several times
    // faster per point than the generic ones, so every problem that is not small runs on
    // them; short needles simply get a longer hop.
    if (span > (1ll << 19)) {
        // measured crossover (tools/needle_sweep.py)
        if ((long long)s <= kWideFromSamples) { *logN_out = 21; return AM_OK; }
        if ((long long)s <= kWidestFromSamples) {
            // a short haystack (BASELINE configs[0]: one 60 s window) whose scores fit ONE pair of 2^21
            // blocks does not pay for a pair of 2^22 (half the points, same number of launches)
            long long hop21 = (1ll << 21) - (long long)s + 1;
            if (hop21 >= 8 * kTile) hop21 = (hop21 / kTile) * kTile;
            *logN_out = (hop21 > 0 && out_count <= 2 * hop21) ? 21 : 22;
            return AM_OK;
        }
        if ((long long)s <= kSegmentFrom) {
            // long needles: 2^23, unless the scores fit one pair of 2^22 blocks
            long long hop22 = (1ll << 22) - (long long)s + 1;
            if (hop22 >= 8 * kTile) hop22 = (hop22 / kTile) * kTile;
            *logN_out = (hop22 > 0 && out_count <= 2 * hop22) ? 22 : 23;
            return AM_OK;
        }
    }
    int pref = min_log;
    while (pref < kLogNMax) {
        const double n = (double)((size_t)1 << pref);
        if ((n - (double)s + 1.0) / n >= kMinEfficiency) break;
        ++pref;
    }
    // whole problem in one block if that is smaller
    int single = kLogNMin;
    while (single < kLogNMax && (long long)((size_t)1 << single) < span) ++single;
    *logN_out = std::min(pref, std::max(single, min_log));
    return AM_OK;
}

static int needle_spectrum(am_needle* h, const Plan* pl, const float2** out) {
    Ctx* c = h->ctx;
    const int key = pl->dev.logN;
    auto it = h->spectra.find(key);
    if (it != h->spectra.end()) { *out = it->second; return AM_OK; }
    const size_t N = (size_t)1 << pl->dev.logN;
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);   // (a device-side redo may still read the work matrix)
    int rc = c->work.ensure(std::max<size_t>(N * sizeof(float2), c->work.cap));
    if (rc) return rc;
    float2* hc = nullptr;
    AM_HIP(hipMalloc((void**)&hc, N * sizeof(float2)));
    Job job{};
    job.src = h->d_needle; job.src_len = (long long)h->n; job.lead = 0;
    job.dst = nullptr; job.out_count = 0; job.hop = 1; job.nblocks = 1; job.first_pair = 0;
    hipError_t e;
    {
        ProfScope ps(c, KN_OTHER);
        e = launch_k1(c->stream, job, 1, (float2*)c->work.p, pl->dev);
        if (e == hipSuccess) e = launch_k2_spectrum(c->stream, (float2*)c->work.p, hc, pl->dev);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(hc); return hip_fail(e, "needle spectrum"); }
    h->spectra[key] = hc;
    *out = hc;
    return AM_OK;
}

// half_pipeline = 2: the spectrum as __half2 points times `hscale` (fixed per needle and plan)
static int needle_spectrum16(am_needle* h, const Plan* pl, float hscale, const float2** out) {
    const int key = pl->dev.logN;
    const bool mfma = k2_mfma_enabled() && pl->dev.mf != nullptr && plan_k2_is_r16(pl->dev);   // (the matrix-core row kernel's layout)
    std::map<int, unsigned*>& cache = mfma ? h->spectra16m : h->spectra16;
    auto it = cache.find(key);
    if (it != cache.end()) { *out = reinterpret_cast<const float2*>(it->second); return AM_OK; }
    const float2* hc = nullptr;
    int rc = needle_spectrum(h, pl, &hc);
    if (rc) return rc;
    Ctx* c = h->ctx;
    const size_t N = (size_t)1 << pl->dev.logN;
    unsigned* h16 = nullptr;
    AM_HIP(hipMalloc((void**)&h16, N * sizeof(unsigned)));
    hipError_t e;
    { ProfScope ps(c, KN_OTHER);
      e = mfma ? launch_spectrum_to_half_mfma(c->stream, hc, (long long)N, hscale, h16) : launch_spectrum_to_half(c->stream, hc, (long long)N, hscale, h16); }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(h16); return hip_fail(e, "needle spectrum (f16)"); }
    cache[key] = h16;
    *out = reinterpret_cast<const float2*>(h16);
    return AM_OK;
}

// The overlap-save engine: scores[j] = factor * sum_n X[j + n - lead] needle[n]
// When a ScanRequest is given and the plan supports it, K3 also writes the level-0
// (min,max) summary into the chosen set's stats32 and `fused` becomes true.
struct ScanRequest {
    float margin;            // in: a run's raw scores are written when its maximum reaches min(its K3 tile's minimum, hist_min) + margin; < 0: all
    float hist_min;          // in: lowest chunk minimum of the needle's recent haystacks (FLT_MAX: none)
    long long seg_c, seg_d;  // in: chunk geometry (scores i*seg_c .. i*seg_c + seg_d)
    int set;                 // in: which set of score-side buffers (0, or 1 in an overlapped batch)
    hipEvent_t before_k3;    // in: K3 must not overwrite that set before this event (or null)
    // in: restrict the launch to the blocks that produce scores [range_a, range_b) (range_b = 0:
    // everything).  Used to redo single chunks with theta = -inf in place.
    long long range_a, range_b;
    int* bad;                // in: host-visible word the summary kernels of the pick set when a score is not finite, or null
    // in (streaming ingest): summary / flag buffers owned by the caller instead of the context's sets, and
    // "launch nothing" (every pair was computed while the samples arrived; only describe what is there)
    DevBuf* ext_stats32; DevBuf* ext_side;
    // in (streaming ingest): the block count the side buffer is laid out for (0: this launch's own).  The
    // thresholds sit behind the ballots, i.e. at an offset that depends on the block count: early pairs are
    // launched under the layout of the announced length and the final pass must keep that layout even
    // when the real length gives fewer blocks.
    long long side_nblocks;
    bool skip_launch;
    bool tail_by_caller;     // in: the caller computes a TailPlan's scores itself (match_many, several haystacks per launch): main pass only
    bool no_scan;            // in: only the block restriction (range_a, range_b) applies; K3 writes plain scores
    bool work_by_set;        // in: the work matrix of set 1 is the context's second one (kept for a device-side redo)
    // out: what a second K3 launch over the same work matrix needs (valid when redo_ok)
    bool redo_ok;
    Job redo_job; PlanDev redo_pl; float redo_scale; int redo_half; int redo_npairs; const float2* redo_work; ScanCfg redo_cfg;
    bool fused;              // out: K3 produced stats32 / wflags
    SparseScores sparse;     // out: description of what was written
};
struct Geometry {
    int logN;
    long long N, hop, nblocks, npairs;
};
static int plan_geometry(size_t s, long long out_count, const Opts& o, Geometry* g) {
    int rc = pick_log_n(s, out_count, o, &g->logN);
    if (rc) return rc;
    g->N = 1ll << g->logN;
    g->hop = g->N - (long long)s + 1;
    if (g->hop >= 8 * kTile) g->hop = (g->hop / kTile) * kTile;
    g->nblocks = (out_count + g->hop - 1) / g->hop;
    g->npairs = (g->nblocks + 1) / 2;
    return AM_OK;
}
// The odd last block.  Two blocks share one complex transform, so a haystack with an odd number of blocks pays a
// whole pair for its last, usually part-filled block (1 h at 44.1 kHz against a 10 s needle: 42.2 blocks of the
// 2^22 plan = 22 pairs, 2.3 % of the points for nothing).  When the scores behind the last even block boundary T fit
// into one pair of a smaller plan that has the fused scan, the main pass stops at T and those scores come from
// that plan, computed on a stream of their own beside the main pass (run_tail_block): every run written, and the
// main layout's ballots / thresholds of the block they belong to preset to "all written", so that the peak pick
// sees one score array with one geometry.  Which blocks a haystack gets depends on its own length only: its bits
// do not depend on the batch it travels in.
struct TailPlan {
    bool on;
    long long T;      // first score of the tail (a multiple of the main plan's hop, hence of kTile)
    Geometry g;       // the smaller plan's layout for scores [T, out_count): one pair
};
static bool tail_plan(size_t s, long long out_count, const Opts& o, const Geometry& g, TailPlan* t) {
    t->on = false;
    if (!o.tail_block || o.log_n != 0 || g.logN < 22 || !(g.nblocks & 1) || g.nblocks < 3 || (g.hop % kTile) != 0) return false;
    const long long T = (g.nblocks - 1) * g.hop, rest = out_count - T;
    for (int lt = 21; lt < g.logN; ++lt) {   // (2^21 is the smallest plan whose K3 carries the scan)
        const long long N = 1ll << lt;
        long long hop = N - (long long)s + 1;
        if (hop < 8 * kTile) continue;
        hop = (hop / kTile) * kTile;
        if (2 * hop < rest) continue;
        t->on = true; t->T = T;
        t->g.logN = lt; t->g.N = N; t->g.hop = hop; t->g.nblocks = (rest + hop - 1) / hop; t->g.npairs = 1;
        return true;
    }
    return false;
}

// Half-precision levels (option "half_pipeline"): 1 = the work matrix travels through HBM as f16,
// butterflies in f32; 2 = K2's butterflies in packed f16 as well.  The scales keep every stored
// or f16-computed value inside f16's range: level 1 normalises K2's product by the needle energy
// (times a fixed gain); level 2 scales the row by 2^-7 on the way into K2 (a full-scale tone then
// peaks at 2^15 in the forward spectrum) and the needle spectrum to an rms of 1/8 per bin.  K3
// divides the scales out in f32.
struct HalfScale {
    int level;
    float pre, hscale;
    float k3(float factor) const { return level ? factor / (hscale * pre) : factor; }
};
static HalfScale half_scale(const am_needle* h, const Opts& o, const PlanDev& pl) {
    HalfScale s{0, 1.0f, 1.0f};
    if (!o.half || !(plan_is_r16(pl) || plan_is_c512(pl))) return s;
    s.level = o.half >= 2 ? 2 : 1;
    if (s.level == 1) s.hscale = kHalfGain * h->inv_autocorr;
    else {
        s.pre = 1.0f / 128.0f;
        s.hscale = (float)((double)(1ull << pl.logN) * std::sqrt((double)h->inv_autocorr) / 8.0);
    }
    return s;
}

// Layout of a set's sparse-score side buffer: the ballots of K3's wavefronts (one 64-bit word per
// block, column tile and wavefront: which of the tile's runs were written), then the write
// thresholds K3 used, one float per (block, column tile).
static size_t sparse_word_bytes(long long nblocks, const PlanDev& pl) {
    return sizeof(unsigned long long) * (((size_t)nblocks << (pl.logN2 - kColsLog)) << (pl.logN1 - 6));
}
static size_t sparse_bytes(long long nblocks, const PlanDev& pl) {
    return sparse_word_bytes(nblocks, pl) + sizeof(float) * ((size_t)nblocks << (pl.logN2 - kColsLog));
}
static void fill_scan_cfg(ScanCfg* cfg, void* stats32, void* side, long long nblocks, const PlanDev& pl, float margin, float hist_min,
                          long long seg_c, long long seg_d) {
    cfg->stats32 = static_cast<float2*>(stats32);
    cfg->wbits = static_cast<unsigned long long*>(side);
    cfg->tile_theta = reinterpret_cast<float*>(static_cast<char*>(side) + sparse_word_bytes(nblocks, pl));
    cfg->margin = margin;
    cfg->hist_min = hist_min;
    cfg->seg_c = seg_c; cfg->seg_d = seg_d;
    cfg->inv_c = seg_c > 0 ? 1.0 / (double)seg_c : 0.0;
}
// what the peak pick sees of it: with every run written (margin < 0) it needs neither flags nor thresholds
static SparseScores sparse_view(const ScanCfg& cfg, long long hop, const PlanDev& pl) {
    if (cfg.margin < 0.0f) return SparseScores{nullptr, cfg.stats32, nullptr, (int)hop, pl.logN2, pl.logN1, 1.0 / (double)hop};
    return SparseScores{cfg.wbits, cfg.stats32, cfg.tile_theta, (int)hop, pl.logN2, pl.logN1, 1.0 / (double)hop};
}

static bool needle_is_segmented(const am_needle* h, const Opts& o) {
    return (long long)h->n > kSegmentFrom && o.log_n == 0;
}
static int needle_segments(am_needle* h) {
    if (!h->segments.empty()) return AM_OK;
    const long long n = (long long)h->n;
    const long long nseg = (n + kSegmentLen - 1) / kSegmentLen;
    for (long long i = 0; i < nseg; ++i) {
        const long long a = n * i / nseg, b = n * (i + 1) / nseg;
        am_needle* sub = new am_needle();
        sub->ctx = h->ctx; sub->d_needle = h->d_needle + a; sub->n = (size_t)(b - a);
        sub->inv_autocorr = h->inv_autocorr; sub->owns_data = false;
        h->segments.push_back(sub);
        h->seg_off.push_back(a);
    }
    return AM_OK;
}

static int run_correlation_one(am_needle* h, const Opts& o, const void* d_src, long long src_len, long long lead,
                               float* d_dst, long long out_count, float factor,
                               ScanRequest* scan_req, int src_kind, bool accumulate);

// The scores [tail.T, out_count) of a haystack on the smaller plan (TailPlan), queued on the context's tail stream:
// one block pair through K1 / K2 / K3 with every run written and the level-0 summary at its place in the main
// pass's stats32; then block `main_nblocks - 1` of the MAIN layout is marked "every run written, threshold -inf".
static int run_tail_block(am_needle* h, const Opts& o, const TailPlan& tail, const void* d_src, long long src_len,
                          float* d_dst, long long out_count, float factor, const ScanCfg& main_scan, const PlanDev& main_pl,
                          long long main_nblocks, int src_kind) {
    Ctx* c = h->ctx;
    hipStream_t st = c->stream_tail;
    int rc;
    const Plan* pl = nullptr;
    if ((rc = get_plan(c, tail.g.logN, &pl))) return rc;
    const float2* hc = nullptr;
    if ((rc = needle_spectrum(h, pl, &hc))) return rc;
    const HalfScale hs = half_scale(h, o, pl->dev);
    if (hs.level == 2 && (rc = needle_spectrum16(h, pl, hs.hscale, &hc))) return rc;
    if ((rc = c->work_tail.ensure((size_t)tail.g.N * sizeof(float2)))) return rc;
    Job job{};
    job.src = static_cast<const char*>(d_src) + 4 * (size_t)tail.T;   // (one f32 sample and one i16 stereo frame are both 4 bytes)
    job.src_len = src_len - tail.T; job.lead = 0; job.src_kind = src_kind;
    job.dst = d_dst + tail.T; job.out_count = out_count - tail.T; job.hop = (int)tail.g.hop; job.nblocks = (int)tail.g.nblocks;
    job.first_pair = 0;
    ScanCfg scan{};
    scan.stats32 = main_scan.stats32 ? main_scan.stats32 + tail.T / 32 : nullptr;
    scan.margin = -1.0f; scan.hist_min = FLT_MAX;
    // (profiled as "other": the three classes' figures stay those of the main pass's launches)
    { ProfScope ps(c, KN_OTHER, st); AM_HIP(launch_k1(st, job, 1, (float2*)c->work_tail.p, pl->dev, hs.level)); }
    { ProfScope ps(c, KN_OTHER, st); AM_HIP(launch_k2(st, 1, (float2*)c->work_tail.p, hc, pl->dev, nullptr, hs.level, hs.hscale, hs.pre, true)); }
    { ProfScope ps(c, KN_OTHER, st); AM_HIP(launch_k3(st, job, 1, (const float2*)c->work_tail.p, pl->dev, hs.k3(factor), scan, hs.level, false)); }
    if (main_scan.stats32 && main_scan.margin >= 0.0f && main_scan.wbits && main_scan.tile_theta) {
        const size_t tiles = (size_t)1 << (main_pl.logN2 - kColsLog), words = tiles << (main_pl.logN1 - 6);
        const size_t blk = (size_t)(main_nblocks - 1);
        AM_HIP(hipMemsetAsync(main_scan.wbits + blk * words, 0xFF, words * sizeof(unsigned long long), st));
        AM_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(main_scan.tile_theta + blk * tiles), (int)0xFF7FFFFFu, tiles, st));   // -FLT_MAX
    }
    return AM_OK;
}

// The overlap-save engine for any needle length: one pass, or one pass per needle segment with the
// source shifted by the segment's offset and K3 adding up the partial sums (plain scores, every one
// written; the peak pick summarises them with tile_stats instead of the fused scan).
static int run_correlation(am_needle* h, const Opts& o, const void* d_src, long long src_len, long long lead,
                           float* d_dst, long long out_count, float factor,
                           ScanRequest* scan_req = nullptr, int src_kind = 0) {
    if (!needle_is_segmented(h, o)) return run_correlation_one(h, o, d_src, src_len, lead, d_dst, out_count, factor, scan_req, src_kind, false);
    int rc = needle_segments(h);
    if (rc) return rc;
    if (scan_req && scan_req->skip_launch) return fail(AM_ERR_INVALID_ARG, "internal: streaming ingest does not run early pairs for partitioned needles");
    Opts os = o;
    os.half = 0;   // (the accumulating K3 exists for the f32 work matrix)
    const size_t nseg = h->segments.size();
    for (size_t i = 0; i < nseg; ++i) {
        // every pass writes (i = 0) or adds (i > 0) plain scores; a pass still honours the restriction to the
        // blocks of one chunk, and the first one may not touch the score buffer before the pick that last
        // read it is done
        ScanRequest plain{};
        ScanRequest* sr = nullptr;
        if (scan_req) {
            plain.margin = -1.0f; plain.range_a = scan_req->range_a; plain.range_b = scan_req->range_b;
            plain.before_k3 = i == 0 ? scan_req->before_k3 : nullptr;
            plain.no_scan = true;
            sr = &plain;
        }
        if ((rc = run_correlation_one(h->segments[i], os, d_src, src_len, lead - h->seg_off[i], d_dst, out_count, factor, sr, src_kind, i > 0)))
            return rc;
    }
    if (scan_req) {   // the sums are complete scores without a level-0 summary: the pick summarises them itself (tile_stats)
        scan_req->fused = false;
        scan_req->sparse = SparseScores{nullptr, nullptr, nullptr, 1, 5, 5, 1.0};
    }
    return AM_OK;
}

static int run_correlation_one(am_needle* h, const Opts& o, const void* d_src, long long src_len, long long lead,
                               float* d_dst, long long out_count, float factor,
                               ScanRequest* scan_req, int src_kind, bool accumulate) {
    Ctx* c = h->ctx;
    if (h->n <= (size_t)kDirectMaxNeedle && o.log_n == 0) {
        // tiny needle: direct summation, every score written, no fused scan
        if (scan_req) {
            scan_req->fused = false;
            scan_req->sparse = SparseScores{nullptr, nullptr, nullptr, 1, 5, 5, 1.0};
        }
        Job job{};
        job.src = d_src; job.src_len = src_len; job.lead = lead; job.src_kind = src_kind;
        job.dst = d_dst; job.out_count = out_count;
        ProfScope ps(c, KN_OTHER);
        AM_HIP(launch_direct(c->stream, job, h->d_needle, (int)h->n, factor));
        return AM_OK;
    }
    Geometry g{};
    int rc = plan_geometry(h->n, out_count, o, &g);
    if (rc) return rc;
    const Plan* pl = nullptr;
    if ((rc = get_plan(c, g.logN, &pl))) return rc;
    const float2* hc = nullptr;
    if ((rc = needle_spectrum(h, pl, &hc))) return rc;
    const long long N = g.N, hop = g.hop, nblocks = g.nblocks;
    // (streaming ingest launches its pairs itself, under the layout of the announced length: no tail there)
    TailPlan tail{};
    if (scan_req && !scan_req->no_scan && !accumulate && lead == 0 && !scan_req->ext_stats32 && !scan_req->ext_side &&
        !scan_req->skip_launch && scan_req->side_nblocks == 0 && plan_has_scan(pl->dev) && c->stream_tail && c->ev_fork && c->ev_join)
        tail_plan(h->n, out_count, o, g, &tail);
    const long long npairs = tail.on ? g.npairs - 1 : g.npairs;   // block pairs of the main pass
    long long ppg = std::max<long long>(1, o.pairs_per_group);
    if (ppg > npairs) ppg = npairs;
    DevBuf& wk = (scan_req && scan_req->work_by_set && scan_req->set) ? c->work_b : c->work;
    if ((rc = wk.ensure((size_t)ppg * (size_t)N * sizeof(float2)))) return rc;
    if (scan_req) scan_req->redo_ok = false;
    ScanCfg scan{};
    if (scan_req && !scan_req->no_scan) {
        scan_req->fused = false;
        scan_req->sparse = SparseScores{nullptr, nullptr, nullptr, (int)hop, pl->dev.logN2, pl->dev.logN1, 1.0 / (double)hop};
        if (plan_has_scan(pl->dev) && (hop % kTile) == 0) {
            DevBuf& b32 = scan_req->ext_stats32 ? *scan_req->ext_stats32 : (scan_req->set ? c->stats32_b : c->stats32);
            DevBuf& bwf = scan_req->ext_side ? *scan_req->ext_side : (scan_req->set ? c->wflags_b : c->wflags);
            const long long side_blocks = std::max(nblocks, scan_req->side_nblocks);
            if ((rc = b32.ensure((size_t)((out_count + 31) / 32) * sizeof(float2)))) return rc;
            if ((rc = bwf.ensure(sparse_bytes(side_blocks, pl->dev)))) return rc;
            fill_scan_cfg(&scan, b32.p, bwf.p, side_blocks, pl->dev, scan_req->margin, scan_req->hist_min, scan_req->seg_c, scan_req->seg_d);
            scan_req->fused = true;
            scan_req->sparse = sparse_view(scan, hop, pl->dev);
        }
    }
    // half-precision storage of the work matrix: K2 normalises by the needle
    // energy (times a fixed gain) so that stored values sit mid-range in f16
    const HalfScale hs = half_scale(h, o, pl->dev);
    const float k3scale = hs.k3(factor);
    if (hs.level == 2 && (rc = needle_spectrum16(h, pl, hs.hscale, &hc))) return rc;
    Job job{};
    job.src = d_src; job.src_len = src_len; job.lead = lead; job.src_kind = src_kind;
    job.dst = d_dst; job.out_count = tail.on ? tail.T : out_count; job.hop = (int)hop; job.nblocks = (int)(tail.on ? nblocks - 1 : nblocks);
    if (scan_req && scan_req->skip_launch) return AM_OK;
    long long pair_lo = 0, pair_hi = npairs;
    bool with_tail = tail.on;
    if (scan_req && scan_req->range_b > scan_req->range_a) {
        pair_lo = (scan_req->range_a / hop) / 2;
        pair_hi = std::min(npairs, ((scan_req->range_b - 1) / hop) / 2 + 1);
        with_tail = tail.on && scan_req->range_b > tail.T;
    }
    if (scan_req && scan_req->tail_by_caller) with_tail = false;
    if (with_tail) {
        // beside the main pass: everything this stream has been told to wait for (the pick that last read the set)
        // holds for the tail's stream too, and the main stream takes the tail back in before anything reads the scores
        AM_HIP(hipEventRecord(c->ev_fork, c->stream));
        AM_HIP(hipStreamWaitEvent(c->stream_tail, c->ev_fork, 0));
        if ((rc = run_tail_block(h, o, tail, d_src, src_len, d_dst, out_count, factor, scan, pl->dev, nblocks, src_kind))) return rc;
        AM_HIP(hipEventRecord(c->ev_join, c->stream_tail));
    }
    bool waited = false;
    for (long long first = pair_lo; first < pair_hi; first += ppg) {
        const int np = (int)std::min(ppg, pair_hi - first);
        job.first_pair = (int)first;
        { ProfScope ps(c, KN_K1); AM_HIP(launch_k1(c->stream, job, np, (float2*)wk.p, pl->dev, hs.level)); }
        { ProfScope ps(c, KN_K2); AM_HIP(launch_k2(c->stream, np, (float2*)wk.p, hc, pl->dev, nullptr, hs.level, hs.hscale, hs.pre)); }
        if (!waited && scan_req && scan_req->before_k3) AM_HIP(hipStreamWaitEvent(c->stream, scan_req->before_k3, 0));
        waited = true;
        { ProfScope ps(c, KN_K3); AM_HIP(launch_k3(c->stream, job, np, (const float2*)wk.p, pl->dev, k3scale, scan, hs.level, accumulate)); }
    }
    if (with_tail) AM_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    if (scan_req && scan_req->fused && !accumulate && pair_lo == 0 && pair_hi == npairs && npairs <= ppg) {
        // the whole haystack's inverse rows sit in one work matrix: K3 can run again over chosen pairs
        scan_req->redo_ok = true;
        job.first_pair = 0;
        scan_req->redo_job = job; scan_req->redo_pl = pl->dev; scan_req->redo_scale = k3scale; scan_req->redo_half = hs.level;
        scan_req->redo_npairs = (int)npairs; scan_req->redo_work = (const float2*)wk.p; scan_req->redo_cfg = scan;
    }
    return AM_OK;
}

// What the transforms of one haystack need of the context's scratch buffers, so that a batch can size them
// once, for its largest haystack, before anything is queued (see QueueingScope).  Also builds the plan and
// the needle spectrum the haystack will use (building one runs kernels and waits for them).
struct Footprint {
    size_t work = 0, stats32 = 0, side = 0, work_tail = 0;
    long long npairs = 0;
    void take(const Footprint& f) {
        work = std::max(work, f.work); stats32 = std::max(stats32, f.stats32); side = std::max(side, f.side);
        work_tail = std::max(work_tail, f.work_tail);
        npairs = std::max(npairs, f.npairs);
    }
};
static int correlation_footprint(am_needle* h, const Opts& o, long long out_count, Footprint* f) {
    if (needle_is_segmented(h, o)) {
        int rc = needle_segments(h);
        if (rc) return rc;
        Opts os = o;
        os.half = 0;
        for (am_needle* sub : h->segments) {
            Footprint one;
            if ((rc = correlation_footprint(sub, os, out_count, &one))) return rc;
            f->work = std::max(f->work, one.work);   // (plain scores: no summary, no flags)
        }
        return AM_OK;
    }
    if (h->n <= (size_t)kDirectMaxNeedle && o.log_n == 0) return AM_OK;
    Geometry g{};
    int rc = plan_geometry(h->n, out_count, o, &g);
    if (rc) return rc;
    const Plan* pl = nullptr;
    if ((rc = get_plan(h->ctx, g.logN, &pl))) return rc;
    const float2* hc = nullptr;
    if ((rc = needle_spectrum(h, pl, &hc))) return rc;
    const HalfScale hs = half_scale(h, o, pl->dev);
    if (hs.level == 2 && (rc = needle_spectrum16(h, pl, hs.hscale, &hc))) return rc;
    const long long ppg = std::min(std::max<long long>(1, o.pairs_per_group), g.npairs);
    f->work = std::max(f->work, (size_t)ppg * (size_t)g.N * sizeof(float2));
    f->npairs = std::max(f->npairs, g.npairs);
    if (plan_has_scan(pl->dev) && (g.hop % kTile) == 0) {
        f->stats32 = std::max(f->stats32, (size_t)((out_count + 31) / 32) * sizeof(float2));
        f->side = std::max(f->side, sparse_bytes(g.nblocks, pl->dev));
        TailPlan tail{};
        if (tail_plan(h->n, out_count, o, g, &tail)) {   // (plan and spectrum of the odd last block's transform, see run_tail_block)
            const Plan* plt = nullptr;
            if ((rc = get_plan(h->ctx, tail.g.logN, &plt))) return rc;
            if ((rc = needle_spectrum(h, plt, &hc))) return rc;
            const HalfScale hst = half_scale(h, o, plt->dev);
            if (hst.level == 2 && (rc = needle_spectrum16(h, plt, hst.hscale, &hc))) return rc;
            f->work_tail = std::max(f->work_tail, (size_t)tail.g.N * sizeof(float2));
        }
    }
    return AM_OK;
}

static float scale_factor(const am_needle* h, int scale, size_t w) {
    if (scale == AM_SCALE_LIB) return h->inv_autocorr;                 // audio_matcher.rs:306-308
    if (scale == AM_SCALE_MY) return h->inv_autocorr / (float)w;       // audio_matcher.rs:444-447
    return 1.0f;
}

static size_t mode_len(size_t w, size_t s, int mode) {                // audio_matcher.rs:450-456
    if (mode == AM_MODE_FULL) return w + s - 1;
    if (mode == AM_MODE_SAME) return w;
    return (w > s ? w - s : 0) + 1;
}

// Duration::from_secs_f64(start as f64 / sr as f64) in whole nanoseconds
// (matcher/mod.rs:127-129); exact on the f64 bits, round-to-nearest-even.
static uint64_t start_nanos(uint64_t start, uint32_t sr) {
    const double t = (double)start / (double)sr;
    if (!(t > 0.0)) return 0;
    int e = 0;
    const double m = std::frexp(t, &e);
    const unsigned long long mant = (unsigned long long)std::ldexp(m, 53);
    const int sh = e - 53;
    unsigned __int128 v = (unsigned __int128)mant * 1000000000ull;
    if (sh >= 0) return (uint64_t)(v << sh);
    const int r = -sh;
    if (r >= 127) return 0;
    unsigned __int128 q = v >> r;
    const unsigned __int128 rem = v & (((unsigned __int128)1 << r) - 1);
    const unsigned __int128 half = (unsigned __int128)1 << (r - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    return (uint64_t)q;
}

// audio_matcher.rs:143-160
static bool is_overshadowed(const am_peak& element, const am_peak* other, uint32_t sr, double max_distance_s) {
    if (!other) return false;
    uint64_t e = start_nanos(element.start, sr), b = start_nanos(other->start, sr);
    if (e < b) std::swap(e, b);
    const uint64_t maxd = (uint64_t)std::llround(max_distance_s * 1e9);
    return (e - b) < maxd && other->prominence > element.prominence;
}

// Makes `segs` the chunk list resident on the device.  Consecutive calls with
// the same geometry (the common case: many haystacks of one length) reuse it.
static int upload_segments(Ctx* c, const std::vector<Segment>& segs) {
    const size_t bytes = sizeof(Segment) * segs.size();
    if (c->segs.p && segs.size() == c->segs_resident.size() &&
        memcmp(segs.data(), c->segs_resident.data(), bytes) == 0)
        return AM_OK;
    int rc;
    c->segs_resident.clear();
    if ((rc = c->segs.ensure(bytes))) return rc;
    if ((rc = c->pinned.ensure(bytes))) return rc;
    memcpy(c->pinned.p, segs.data(), bytes);
    AM_HIP(hipMemcpyAsync(c->segs.p, c->pinned.p, bytes, hipMemcpyHostToDevice, c->stream));
    // the staging buffer is reused by the next upload: finish this one first (rare path)
    AM_HIP(hipStreamSynchronize(c->stream));
    c->segs_resident = segs;
    return AM_OK;
}

// The result area of one call, in coherent pinned host memory that the peak kernel
// writes directly: `nhdr` per-chunk headers followed by a spill arena for the peak
// lists of chunks with more than kInlinePeaks peaks (a bump allocator in the
// kernel; its cursor lives in device memory and is zeroed per call).
static int prepare_results(Ctx* c, size_t nhdr, size_t arena_entries, PeakArena* arena) {
    const size_t hdr_bytes = (sizeof(SegHeader) * nhdr + 63) / 64 * 64;
    int rc;
    if ((rc = c->hdr.ensure(hdr_bytes + sizeof(am_peak) * arena_entries))) return rc;
    if ((rc = c->arena_cur.ensure(sizeof(unsigned)))) return rc;
    AM_HIP(hipMemsetAsync(c->arena_cur.p, 0, sizeof(unsigned), c->stream));
    arena->base = reinterpret_cast<am_peak*>(static_cast<char*>(c->hdr.p) + hdr_bytes);
    arena->cursor = static_cast<unsigned*>(c->arena_cur.p);
    arena->cap = (unsigned)arena_entries;
    return AM_OK;
}

// Launches find_peaks (audio_matcher.rs:221-230) for `nsegs` segments of a
// resident score array; segment descriptors live at [seg_off, seg_off + nsegs) of the
// context's segment buffer, result headers at [hdr_off, hdr_off + nsegs).
static int launch_pick(Ctx* c, const float* d_scores, long long n_scores, int seg_off, int nsegs,
                       float min_prom, long long min_dist, const ScanRequest* scan, int hdr_off,
                       const PeakArena& arena, const PeakPolicy& pol, hipStream_t st = nullptr, bool only_failed = false) {
    if (!st) st = c->stream;
    const int set = scan ? scan->set : 0;
    DevBuf& bstats = set ? c->stats_b : c->stats;
    DevBuf& bpeaks = set ? c->peaks_b : c->peaks;
    const float2* d_stats32 = (scan && scan->fused) ? scan->sparse.stats32 : nullptr;
    const SparseScores sp = (scan && scan->fused) ? scan->sparse : SparseScores{nullptr, nullptr, nullptr, 1, 5, 5, 1.0};
    if (nsegs == 0 || n_scores <= 0) return AM_OK;
    int rc;
    const long long ntiles = (n_scores + kTile - 1) / kTile;
    if ((rc = bstats.ensure((size_t)ntiles * sizeof(float2)))) return rc;
    if (!only_failed) {   // (a second pick after a device-side redo of K3 finds the summaries it left: the scores are the same)
        ProfScope ps(c, KN_STATS, st);
        int* bad = scan ? scan->bad : nullptr;
        if (d_stats32) AM_HIP(launch_stats_reduce(st, d_stats32, n_scores, (float2*)bstats.p, bad));
        else AM_HIP(launch_tile_stats(st, d_scores, n_scores, (float2*)bstats.p, bad));
    }
    // hand-over area for chunks with many candidate tiles (per chunk of this launch; the picks
    // of one call run in stream order, so one area serves them all)
    if ((rc = c->wide_ctl.ensure((size_t)nsegs * 24))) return rc;
    if ((rc = c->wide_list.ensure((size_t)nsegs * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
    WideState wide{};
    wide.best = static_cast<unsigned long long*>(c->wide_ctl.p);
    wide.state = reinterpret_cast<int*>(wide.best + nsegs);
    wide.count = reinterpret_cast<unsigned*>(wide.state + nsegs);
    wide.seg_min = reinterpret_cast<float*>(wide.state + 2 * nsegs);
    wide.ntiles = wide.state + 3 * nsegs;
    if ((rc = c->wide_tiles.ensure((size_t)nsegs * kWideTileList * sizeof(int)))) return rc;
    wide.tiles = static_cast<int*>(c->wide_tiles.p);
    wide.list = static_cast<am_peak*>(c->wide_list.p);
    wide.cap = AM_MAX_PEAKS_PER_CHUNK;
    {
        ProfScope ps(c, KN_PEAKS, st);
        AM_HIP(launch_peaks(st, d_scores, n_scores, (const float2*)bstats.p,
                            (const Segment*)c->segs.p + seg_off, nsegs, min_prom, min_dist,
                            (am_peak*)bpeaks.p, (SegHeader*)c->hdr.p + hdr_off, sp, arena, wide, only_failed, pol));
    }
    return AM_OK;
}

// The picks of a needle group (several needles against one haystack) as ONE set of launches: the level-1 summaries, the
// per-chunk pick and its two follow-up kernels each run once with the needle on a grid dimension, instead of four small
// launches per needle.  Every needle's result headers go to hdr_off[z] (absolute); scratch is laid out needle after needle.
static int launch_pick_group(Ctx* c, const K3Group& kg, long long n_scores, int seg_off, int nsegs, float min_prom, long long min_dist,
                             const SparseScores& sp_common, int* bad, const int* hdr_off, const PeakArena& arena, const PeakPolicy& pol,
                             hipStream_t st) {
    if (nsegs == 0 || n_scores <= 0 || kg.n <= 0) return AM_OK;
    int rc;
    const size_t nz = (size_t)kg.n, total = nz * (size_t)nsegs;
    const long long ntiles = (n_scores + kTile - 1) / kTile;
    PickGroup pg{};
    pg.n = kg.n;
    for (int z = 0; z < kg.n; ++z) {
        if ((rc = c->grp_stats[z].ensure((size_t)ntiles * sizeof(float2)))) return rc;
        pg.g[z] = kg.dst[z]; pg.stats[z] = static_cast<float2*>(c->grp_stats[z].p);
        pg.stats32[z] = kg.stats32[z]; pg.wbits[z] = kg.wbits[z]; pg.theta[z] = kg.tile_theta[z];
        pg.hdr_off[z] = hdr_off[z];
    }
    { ProfScope ps(c, KN_STATS, st);
      AM_HIP(launch_stats_reduce(st, pg.stats32[0], n_scores, pg.stats[0], bad, &pg)); }
    if ((rc = c->wide_ctl.ensure(total * 24))) return rc;
    if ((rc = c->wide_list.ensure(total * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
    if ((rc = c->wide_tiles.ensure(total * kWideTileList * sizeof(int)))) return rc;
    if ((rc = c->peaks.ensure(total * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
    WideState wide{};
    wide.best = static_cast<unsigned long long*>(c->wide_ctl.p);
    wide.state = reinterpret_cast<int*>(wide.best + total);
    wide.count = reinterpret_cast<unsigned*>(wide.state + total);
    wide.seg_min = reinterpret_cast<float*>(wide.state + 2 * total);
    wide.ntiles = wide.state + 3 * total;
    wide.tiles = static_cast<int*>(c->wide_tiles.p);
    wide.list = static_cast<am_peak*>(c->wide_list.p);
    wide.cap = AM_MAX_PEAKS_PER_CHUNK;
    {
        ProfScope ps(c, KN_PEAKS, st);
        AM_HIP(launch_peaks(st, pg.g[0], n_scores, pg.stats[0], (const Segment*)c->segs.p + seg_off, nsegs, min_prom, min_dist,
                            (am_peak*)c->peaks.p, (SegHeader*)c->hdr.p, sp_common, arena, wide, false, pol, &pg));
    }
    return AM_OK;
}

// A chunk whose pick reported more than AM_MAX_PEAKS_PER_CHUNK peaks passing the prominence filter
// (SegHeader::overflow & 1): find_peaks returns them all, so does this path.  The scores, their
// tile summary (set 0) and the resident chunk `seg_idx` are those of the pick that just failed.
// Count the qualifying peaks, build the list in global memory, sort and filter it on the device
// (am_peaks.hip, peaks_big_finish), fetch the survivors.  Synchronous; appends to `all`.
static int pick_chunk_big(Ctx* c, const float* d_scores, long long n_scores, int seg_idx, const Segment& sg,
                          float min_prom, long long min_dist, const ScanRequest* scan, float seg_min,
                          std::vector<am_peak>& all, const PeakPolicy& pol) {
    const long long a = sg.a, b = std::min(sg.b, n_scores);
    if (b - a >= 0xFFFFFFFFll) return fail(AM_ERR_PEAK_OVERFLOW, "chunk of 2^32 scores or more with more than AM_MAX_PEAKS_PER_CHUNK peaks");
    const SparseScores sp = (scan && scan->fused) ? scan->sparse : SparseScores{nullptr, nullptr, nullptr, 1, 5, 5, 1.0};
    int rc;
    if ((rc = c->wide_ctl.ensure(24))) return rc;
    struct Ctl { unsigned long long best; int state; unsigned count; float seg_min; int ntiles; } ctl{0ull, 7, 0u, seg_min, -1};   // (state: handed over, head and tail pieces to be scanned)
    WideState wide{};
    wide.best = static_cast<unsigned long long*>(c->wide_ctl.p);
    wide.state = reinterpret_cast<int*>(wide.best + 1);
    wide.count = reinterpret_cast<unsigned*>(wide.state + 1);
    wide.seg_min = reinterpret_cast<float*>(wide.state + 2);
    wide.ntiles = wide.state + 3;
    wide.tiles = nullptr;
    const Segment* d_seg = (const Segment*)c->segs.p + seg_idx;
    // pass 1: count
    wide.list = nullptr; wide.cap = 0;
    AM_HIP(hipMemcpyAsync(c->wide_ctl.p, &ctl, 24, hipMemcpyHostToDevice, c->stream));
    AM_HIP(launch_peaks_wide_one(c->stream, d_scores, n_scores, (const float2*)c->stats.p, d_seg, min_prom, min_dist, sp, wide, pol));
    unsigned n = 0;
    AM_HIP(hipMemcpyAsync(&n, wide.count, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    AM_HIP(hipStreamSynchronize(c->stream));
    if (n == 0) return AM_OK;
    if (n >= 0x40000000u) return fail(AM_ERR_PEAK_OVERFLOW, "peak list build failed");
    // one allocation: list | out | keys (2n) | table | idx (2n) | out_n
    const size_t nb = min_dist > 0 ? (size_t)((b - a) / min_dist) + 3 : 1;
    const size_t off_out = sizeof(am_peak) * (size_t)n, off_keys = 2 * off_out, off_table = off_keys + 16 * (size_t)n,
                 off_idx = off_table + 8 * nb, off_n = off_idx + 8 * (size_t)n;
    if ((rc = c->big.ensure(off_n + 16))) return rc;
    char* base = static_cast<char*>(c->big.p);
    // pass 2: fill the list (in any order)
    wide.list = reinterpret_cast<am_peak*>(base); wide.cap = n;
    AM_HIP(hipMemcpyAsync(c->wide_ctl.p, &ctl, 24, hipMemcpyHostToDevice, c->stream));
    AM_HIP(launch_peaks_wide_one(c->stream, d_scores, n_scores, (const float2*)c->stats.p, d_seg, min_prom, min_dist, sp, wide, pol));
    AM_HIP(hipMemsetAsync(base + off_table, 0xFF, 8 * nb, c->stream));
    AM_HIP(launch_peaks_big_finish(c->stream, wide.list, n, a, min_dist, reinterpret_cast<unsigned long long*>(base + off_keys),
                                   reinterpret_cast<unsigned*>(base + off_idx), reinterpret_cast<long long*>(base + off_table),
                                   reinterpret_cast<am_peak*>(base + off_out), reinterpret_cast<unsigned*>(base + off_n), pol));
    unsigned kept = 0;
    AM_HIP(hipMemcpyAsync(&kept, base + off_n, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    AM_HIP(hipStreamSynchronize(c->stream));
    if (kept > n) return fail(AM_ERR_PEAK_OVERFLOW, "peak filter failed");
    const size_t old = all.size();
    all.resize(old + kept);
    if (kept) {
        AM_HIP(hipMemcpyAsync(all.data() + old, base + off_out, sizeof(am_peak) * (size_t)kept, hipMemcpyDeviceToHost, c->stream));
        AM_HIP(hipStreamSynchronize(c->stream));
    }
    return AM_OK;
}

// windows of common::chunked(chunk + overlap, hop = chunk) (audio_matcher.rs:104)
// as slices of the global score array; a window shorter than the needle has
// no valid lag and is skipped.  `widths` (optional) receives within.len() of each window.
// `drop_tail` (option "tail_window" = 1): chunked() yields full-length windows only.
static void make_segments(size_t len, size_t s, const am_match_params* p, bool drop_tail, std::vector<Segment>& segs,
                          std::vector<size_t>* widths = nullptr, size_t max_windows = (size_t)-1) {
    const unsigned long long window = p->chunk + p->overlap;
    size_t i = 0;
    for (unsigned long long off = 0; off < len && i < max_windows; off += p->chunk, ++i) {
        const unsigned long long w = std::min<unsigned long long>(window, len - off);
        if (w < s || (drop_tail && w < window)) continue;
        Segment sg; sg.a = (long long)off; sg.b = (long long)(off + w - s + 1);
        segs.push_back(sg);
        if (widths) widths->push_back((size_t)w);
    }
}

// sort by start (audio_matcher.rs:135) + filter_surrounding (audio_matcher.rs:136-139)
// `from_filtered` (option "surrounding_from" = 1): the neighbour before an element is the last element that was KEPT (a
// sequential filter); default: both neighbours come from the sorted, unfiltered sequence.
static int merge_peaks(std::vector<am_peak>& all, const am_match_params* p, bool from_filtered, am_peak* out, size_t cap, size_t* n_out) {
    std::stable_sort(all.begin(), all.end(), [](const am_peak& x, const am_peak& y) { return x.start < y.start; });
    size_t n = 0;
    am_peak last_kept{};
    bool have_kept = false;
    for (size_t i = 0; i < all.size(); ++i) {
        const am_peak* before = from_filtered ? (have_kept ? &last_kept : nullptr) : (i > 0 ? &all[i - 1] : nullptr);
        const am_peak* after = i + 1 < all.size() ? &all[i + 1] : nullptr;
        if (is_overshadowed(all[i], before, p->sr, p->overshadow_distance_s) ||
            is_overshadowed(all[i], after, p->sr, p->overshadow_distance_s))
            continue;
        last_kept = all[i]; have_kept = true;
        if (n < cap) out[n] = all[i];
        ++n;
    }
    *n_out = n;
    if (n > cap) return fail(AM_ERR_CAPACITY, "peak output buffer too small");
    return AM_OK;
}

// Appends the peaks of header `hd` (inline, or spilled to the arena) to `all`.
static void append_header_peaks(const SegHeader& hd, const PeakArena& arena, std::vector<am_peak>& all) {
    if (hd.n <= kInlinePeaks) {
        for (int j = 0; j < hd.n; ++j) all.push_back(hd.first[j]);
    } else {
        const am_peak* src = arena.base + hd.arena_off;
        all.insert(all.end(), src, src + hd.n);
    }
}

static inline const void* advance_src(const void* src, size_t elements) {
    // one f32 mono sample and one interleaved i16 stereo frame are both 4 bytes
    return static_cast<const char*>(src) + 4 * elements;
}

// Streaming ingest (am_match_stream_*): the block pairs [0, pairs_done) of the one haystack were
// computed while its samples arrived, into buffers the stream object owns.
// One part of a haystack that is split over several devices (am_match_part_device, am_pool_match_long*): the
// buffer holds the samples from window `first_window` on, only its first `max_windows` windows belong to
// this part (the samples behind them are the last window's overlap), and the peaks come back unmerged, in
// window order, at their positions in the whole haystack -- calc_chunks up to audio_matcher.rs:131.
// Whether the main pass of a haystack with out_count scores leaves its odd last block to a TailPlan -- the conditions
// run_correlation_one applies, for a caller that computes the tails of several haystacks per launch (match_many).
static bool haystack_tail(am_needle* h, const Opts& o, long long out_count, TailPlan* t) {
    t->on = false;
    Ctx* c = h->ctx;
    if (needle_is_segmented(h, o) || (h->n <= (size_t)kDirectMaxNeedle && o.log_n == 0)) return false;
    if (!c->stream_tail || !c->ev_fork || !c->ev_join) return false;
    Geometry g{};
    if (plan_geometry(h->n, out_count, o, &g)) return false;
    return tail_plan(h->n, out_count, o, g, t);   // (main plans of 2^22 points and more: all carry the fused scan)
}
// The tails of up to kMaxTailBatch haystacks of a batch (all on one smaller plan) as ONE launch each of K1 / K2 / K3 on
// the main stream: full grids instead of one under-filled launch triple per haystack beside the main pass (which costs
// about as much as the dropped pair saves, profiles/r04/tail_block_ab.txt).  The scores and their summary go to slots
// of the context's tail buffers; launch_tail_commit moves a haystack's slot into the score-side set its pick reads,
// once the pick that last read that set is done (on the pick's stream).  Same kernels' arithmetic as run_tail_block:
// a haystack's bits do not depend on whether it travels alone or in a batch.
struct TailSlots { size_t scores, stats; };   // elements per slot (floats, float2s)
// (the several-per-launch kernels exist for the 256-row plan, 2^21 points: the tail of a 2^23 main pass that needs the
// 2^22 plan is computed beside its main pass, like a single haystack's)
static bool tail_batchable(const TailPlan& t) { return t.on && t.g.logN == 21; }
static int launch_tail_batch(am_needle* h, const Opts& o, const std::vector<TailPlan>& tails, const std::vector<size_t>& members, int half_idx,
                             const TailSlots& sl, const void* const* d_hays, const size_t* lens, float factor, int src_kind) {
    Ctx* c = h->ctx;
    int rc;
    const TailPlan& first = tails[members[0]];
    const Plan* pl = nullptr;
    if ((rc = get_plan(c, first.g.logN, &pl))) return rc;
    const float2* hc = nullptr;
    if ((rc = needle_spectrum(h, pl, &hc))) return rc;
    const HalfScale hs = half_scale(h, o, pl->dev);
    if (hs.level == 2 && (rc = needle_spectrum16(h, pl, hs.hscale, &hc))) return rc;
    TailBatch tb{};
    tb.n = (int)members.size();
    for (int j = 0; j < tb.n; ++j) {
        const size_t k = members[j];
        const TailPlan& t = tails[k];
        const size_t slot = (size_t)half_idx * kMaxTailBatch + (size_t)j;
        tb.src[j] = static_cast<const char*>(d_hays[k]) + 4 * (size_t)t.T;
        tb.src_len[j] = (long long)lens[k] - t.T;
        tb.out_count[j] = (long long)(lens[k] - h->n + 1) - t.T;
        tb.dst[j] = static_cast<float*>(c->tail_scores.p) + slot * sl.scores;
        tb.stats32[j] = static_cast<float2*>(c->tail_stats.p) + slot * sl.stats;
    }
    float2* work = static_cast<float2*>(c->work_tail.p);
    { ProfScope ps(c, KN_OTHER); AM_HIP(launch_tail_batch_k1(c->stream, tb, (int)first.g.hop, src_kind, work, pl->dev, hs.level)); }
    { ProfScope ps(c, KN_OTHER); AM_HIP(launch_k2(c->stream, tb.n, work, hc, pl->dev, nullptr, hs.level, hs.hscale, hs.pre, true)); }
    { ProfScope ps(c, KN_OTHER); AM_HIP(launch_tail_batch_k3(c->stream, tb, (int)first.g.hop, work, pl->dev, hs.k3(factor), hs.level)); }
    return AM_OK;
}

struct PartSpec {
    size_t max_windows;
    uint64_t first_sample;            // position of the part's first sample in the whole haystack
    size_t chunk_base, chunk_total;   // for the per-chunk progress events: this part's first window, windows of the whole haystack
    std::vector<am_peak>* raw;        // out
};

struct StreamPre {
    float* scores;
    DevBuf* stats32; DevBuf* side;
    long long pairs_done;
    long long layout_nblocks;   // the block count the early pairs laid the side buffer out for (ScanRequest::side_nblocks)
};

// calc_chunks (audio_matcher.rs:88-141) over a batch of resident haystacks =
// the per-file loop of matcher::run (matcher/mod.rs:42-87).  Everything is
// queued on the context's stream without host synchronisation; the per-chunk
// result headers land in pinned host memory, so no copy ends the batch.
//
// scale == AM_SCALE_MY (MyConvolve's semantics, audio_matcher.rs:442-448): the factor
// 1 / (sum(needle^2) * within.len()) depends on the window, so the windows of full
// length share the main pass and every shorter window at the end of a haystack is
// correlated on its own with its own factor.
// Which chunks of a haystack are touched by non-finite samples: drop[i] = the chunk's own window
// holds one (the reference's scores for it are NaN throughout: no peak); again[i] = its window is
// clean but some of its scores came from a block pair that holds one.  One search kernel over the
// sample ranges of all block pairs and all windows; rare path, synchronous.
static int classify_nonfinite(am_needle* h, const Opts& o, const float* d_hay, size_t len, long long out_count,
                              const std::vector<Segment>& segs, int s0, int s1,
                              std::vector<char>* drop, std::vector<char>* again, bool with_tail) {
    Ctx* c = h->ctx;
    const long long s = (long long)h->n;
    const int nch = s1 - s0;
    drop->assign(nch, 0); again->assign(nch, 0);
    std::vector<Segment> ranges;
    for (int i = s0; i < s1; ++i)       // the samples behind scores [a, b): a .. b + s - 2
        ranges.push_back(Segment{segs[i].a, std::min<long long>((long long)len, segs[i].b + s - 1)});
    Geometry g{};
    TailPlan tail{};
    long long npairs = 0;
    const bool segmented = needle_is_segmented(h, o);   // (every segment pass has block pairs of its own: all clean windows again)
    if (!segmented && !(h->n <= (size_t)kDirectMaxNeedle && o.log_n == 0)) {   // (direct summation spreads nothing)
        int rc = plan_geometry(h->n, out_count, o, &g);
        if (rc) return rc;
        npairs = g.npairs;
        // K1 loads a full N samples per block, starting at block * hop (am_fft.hip, k1_cols_fwd_*), and hop
        // may have been rounded down to a multiple of kTile: pair q reads [2q hop, (2q + 1) hop + N), or
        // [2q hop, 2q hop + N) when its second block does not exist -- up to kTile - 1 samples more than
        // the scores it yields depend on, and a NaN there still poisons the whole pair
        if (with_tail) tail_plan(h->n, out_count, o, g, &tail);
        if (tail.on) --npairs;   // (the main pass ends at tail.T; the scores behind it come from one pair of the smaller plan)
        for (long long q = 0; q < npairs; ++q) {
            const long long last_block = (2 * q + 1 < g.nblocks) ? 2 * q + 1 : 2 * q;
            ranges.push_back(Segment{2 * q * g.hop, std::min<long long>((long long)len, last_block * g.hop + g.N)});
        }
        if (tail.on) ranges.push_back(Segment{tail.T, std::min<long long>((long long)len, tail.T + (tail.g.nblocks - 1) * tail.g.hop + tail.g.N)});
    }
    int rc;
    if ((rc = c->ranges.ensure(sizeof(Segment) * ranges.size()))) return rc;
    if ((rc = c->range_flags.ensure(sizeof(int) * ranges.size()))) return rc;
    AM_HIP(hipMemcpyAsync(c->ranges.p, ranges.data(), sizeof(Segment) * ranges.size(), hipMemcpyHostToDevice, c->stream));
    AM_HIP(hipMemsetAsync(c->range_flags.p, 0, sizeof(int) * ranges.size(), c->stream));
    AM_HIP(launch_nonfinite_ranges(c->stream, d_hay, (const Segment*)c->ranges.p, (int)ranges.size(), (int*)c->range_flags.p));
    std::vector<int> flags(ranges.size(), 0);
    AM_HIP(hipMemcpyAsync(flags.data(), c->range_flags.p, sizeof(int) * ranges.size(), hipMemcpyDeviceToHost, c->stream));
    AM_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nch; ++i) {
        if (flags[i]) { (*drop)[i] = 1; continue; }
        if (segmented) { (*again)[i] = 1; continue; }
        const Segment sg = segs[s0 + i];
        for (long long q = 0; q < npairs && !(*again)[i]; ++q)
            if (flags[nch + q] && 2 * q * g.hop < sg.b && (2 * q + 2) * g.hop > sg.a) (*again)[i] = 1;
        if (tail.on && flags[nch + npairs] && tail.T < sg.b) (*again)[i] = 1;
    }
    return AM_OK;
}

static int match_many(am_needle* h, const void* const* d_hays, const size_t* lens, size_t n_hay,
                      const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out, int src_kind = 0,
                      size_t index_base = 0, size_t index_stride = 1, bool fire_hooks = true, const StreamPre* pre = nullptr,
                      const PartSpec* part = nullptr) {
    Ctx* c = h->ctx;
    const Opts o = snapshot_opts(h);
    const PeakPolicy pol = o.peak_policy();
    Hooks hooks = fire_hooks ? snapshot_hooks() : Hooks{};
    if (part) hooks.fn = nullptr;   // (the caller reports the whole haystack; the chunks report themselves, below)
    if (part && n_hay != 1) return fail(AM_ERR_INVALID_ARG, "internal: a part is one haystack");
    // local haystack k is item G(k) of the caller's batch: out, n_out and the progress
    // callbacks use that index (pool submit threads pass their shard: base + k * stride)
    auto G = [&](size_t k) { return index_base + k * index_stride; };
    const size_t s = h->n;
    if (p->chunk == 0) return fail(AM_ERR_INVALID_ARG, "chunk must be > 0");
    if (p->scale < AM_SCALE_NONE || p->scale > AM_SCALE_MY) return fail(AM_ERR_INVALID_ARG, "bad scale");
    const bool my = p->scale == AM_SCALE_MY;
    const size_t window = (size_t)(p->chunk + p->overlap);
    const float factor = scale_factor(h, p->scale, window);
    // Raw scores are written only for the 32-score runs whose maximum reaches their K3 tile's write
    // threshold: the tile's own minimum in the block plus half a prominence (am_fft.hip, k3_finish).
    // The peak kernel certifies per chunk that every threshold was low enough; a chunk that fails
    // (a dip deeper than half a prominence that most tiles' samples missed) is redone with every
    // run written.
    const int sm = p->scale == AM_SCALE_LIB ? 1 : 0;
    const bool sparse_ok = !my && !o.dense && p->min_prominence > 0.f;
    ScanRequest scan{};
    scan.margin = sparse_ok ? 0.5f * p->min_prominence : -1.0f;
    scan.hist_min = h->hist_min(sm);   // (of the haystacks before this call: the whole batch is queued before any result is back)
    scan.seg_c = (long long)p->chunk;
    scan.seg_d = (long long)(p->chunk + p->overlap) - (long long)s;
    // main-pass segments of every haystack, back to back; MyConvolve scaling keeps the
    // shorter windows at the end of a haystack for the second pass
    std::vector<Segment> segs, tail_segs;
    std::vector<size_t> widths, tail_w;
    std::vector<int> seg_off(n_hay + 1, 0), tail_off(n_hay + 1, 0), n_chunks(n_hay, 0);
    size_t max_scores = 0, max_segs = 0;
    for (size_t k = 0; k < n_hay; ++k) {
        n_out[G(k)] = 0;
        seg_off[k] = (int)segs.size();
        tail_off[k] = (int)tail_segs.size();
        if (d_hays[k] && lens[k] >= s) {
            std::vector<Segment> one; std::vector<size_t> w1;
            make_segments(lens[k], s, p, o.tail_window != 0, one, &w1, part ? part->max_windows : (size_t)-1);
            n_chunks[k] = (int)one.size();
            for (size_t i = 0; i < one.size(); ++i) {
                if (my && w1[i] != window) { tail_segs.push_back(one[i]); tail_w.push_back(w1[i]); }
                else { segs.push_back(one[i]); widths.push_back(w1[i]); }
            }
            max_scores = std::max(max_scores, lens[k] - s + 1);
        }
        max_segs = std::max(max_segs, std::max(segs.size() - (size_t)seg_off[k], (size_t)1));
    }
    seg_off[n_hay] = (int)segs.size();
    tail_off[n_hay] = (int)tail_segs.size();
    const size_t nsegs = segs.size();
    if (nsegs == 0 && tail_segs.empty()) return AM_OK;
    if (max_segs > (size_t)1 << 18 || nsegs > (size_t)1 << 24)
        return fail(AM_ERR_INVALID_ARG, "chunk size too small for this haystack (more than 2^18 chunks)");
    int rc;
    // In a batch the peak pick of haystack k (small, latency-bound kernels) runs on a second
    // stream beside the transforms of haystack k+1; the score-side buffers alternate between
    // two sets and K3 waits for the pick that last read the set it is about to overwrite.
    size_t n_active = 0;
    for (size_t k = 0; k < n_hay; ++k) n_active += seg_off[k + 1] > seg_off[k];
    const bool overlap = o.batch_overlap && n_active > 1 && c->stream2 &&
                         c->ev_k3[0] && c->ev_k3[1] && c->ev_pick[0] && c->ev_pick[1];
    if ((rc = c->scores.ensure(max_scores * sizeof(float)))) return rc;
    if ((rc = c->peaks.ensure(sizeof(am_peak) * max_segs * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    // sized once for the longest haystack, so that no pick of the batch has to grow them while
    // the previous pick still runs on the other stream
    if ((rc = c->wide_ctl.ensure(max_segs * 24))) return rc;
    if ((rc = c->wide_list.ensure(max_segs * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
    if ((rc = c->wide_tiles.ensure(max_segs * kWideTileList * sizeof(int)))) return rc;
    if ((rc = c->stats.ensure((max_scores + kTile - 1) / kTile * sizeof(float2)))) return rc;
    if (overlap) {
        if ((rc = c->stats_b.ensure((max_scores + kTile - 1) / kTile * sizeof(float2)))) return rc;
        if ((rc = c->scores_b.ensure(max_scores * sizeof(float)))) return rc;
        if ((rc = c->peaks_b.ensure(sizeof(am_peak) * max_segs * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    }
    // ... and the transforms' own buffers -- work matrix, level-0 summary, ballots and thresholds -- for the
    // haystack that needs the most of each: a ragged batch whose later haystacks are longer must not free
    // and re-allocate them under the kernels of the earlier ones (plans and needle spectra are built here too)
    Footprint need;
    for (size_t k = 0; k < n_hay; ++k) {
        if (n_chunks[k] == 0 || seg_off[k + 1] == seg_off[k]) continue;
        Footprint one;
        if ((rc = correlation_footprint(h, o, (long long)(lens[k] - s + 1), &one))) return rc;
        need.take(one);
    }
    // The odd last blocks (TailPlan).  A single haystack computes its tail beside its main pass (run_tail_block); a
    // batch that overlaps picks and transforms computes the tails of up to kMaxTailBatch haystacks per launch, into
    // slots of two alternating halves (a half is written again two batches later: every commit out of it is long done,
    // the main stream has waited for the pick of the haystack before the previous one by then).
    std::vector<TailPlan> tails(n_hay);
    std::vector<int> tail_slot(n_hay, -1);
    TailSlots tslots{0, 0};
    bool batch_tails = false;
    if (!pre && need.work_tail) {
        size_t n_tails = 0;
        for (size_t k = 0; k < n_hay; ++k) {
            if (n_chunks[k] == 0 || seg_off[k + 1] == seg_off[k]) continue;
            if (haystack_tail(h, o, (long long)(lens[k] - s + 1), &tails[k]) && tail_batchable(tails[k])) {
                ++n_tails;
                tslots.scores = std::max(tslots.scores, (size_t)(2 * tails[k].g.hop));
            }
        }
        tslots.stats = tslots.scores / 32;
        batch_tails = overlap && n_tails > 1;
        const size_t nslot = batch_tails ? kMaxTailBatch : 1;
        if ((rc = c->work_tail.ensure(need.work_tail * nslot))) return rc;
        if (batch_tails) {
            if ((rc = c->tail_scores.ensure(2 * kMaxTailBatch * tslots.scores * sizeof(float)))) return rc;
            if ((rc = c->tail_stats.ensure(2 * kMaxTailBatch * tslots.stats * sizeof(float2)))) return rc;
        }
    }
    int tail_batches = 0;
    for (int set = 0; set < (overlap ? 2 : 1); ++set) {
        if (need.work && (rc = (set ? c->work_b : c->work).ensure(need.work))) return rc;
        if (pre) continue;   // (streaming ingest brings its own summary and flag buffers)
        if (need.stats32 && (rc = (set ? c->stats32_b : c->stats32).ensure(need.stats32))) return rc;
        if (need.side && (rc = (set ? c->wflags_b : c->wflags).ensure(need.side))) return rc;
    }
    // one spare header behind the main ones serves the single-chunk passes below; the arena
    // holds every list of one haystack in the worst case plus a few entries per chunk
    // (bounded: a chunk whose list finds no room is picked again on its own below)
    PeakArena arena{};
    if ((rc = prepare_results(c, nsegs + 1, std::min<size_t>(max_segs * AM_MAX_PEAKS_PER_CHUNK, (size_t)1 << 20) + nsegs * 8, &arena))) return rc;
    // the resident chunk list: the main-pass chunks, then one local slice [0, count) per
    // second-pass window (those are correlated on their own, see below)
    std::vector<Segment> resident = segs;
    for (const Segment& sg : tail_segs) { Segment local; local.a = 0; local.b = sg.b - sg.a; resident.push_back(local); }
    // and one local slice as long as a full chunk, for chunks that are correlated again on their
    // own window (non-finite samples nearby, below); the pick clamps it to the scores there are
    const int local_seg = (int)resident.size();
    // (seg_d is the index of a full window's LAST score: chunk + overlap - s + 1 scores in all)
    { Segment local; local.a = 0; local.b = std::max<long long>(scan.seg_d + 1, 1); resident.push_back(local); }
    if ((rc = upload_segments(c, resident))) return rc;
    if ((rc = c->badflag.ensure(sizeof(int) * n_hay))) return rc;
    int* h_bad = static_cast<int*>(c->badflag.p);
    memset(h_bad, 0, sizeof(int) * n_hay);
    if ((rc = c->failcnt.ensure(nsegs + 1))) return rc;
    unsigned char* h_fail = static_cast<unsigned char*>(c->failcnt.p);
    memset(h_fail, 0, nsegs + 1);
    // A chunk whose certificate fails is redone on the device when the batch overlaps picks and transforms:
    // the pick marks the block pairs that feed it, K3 runs once more for those pairs with every run written
    // (from the haystack's own work matrix: two alternate) and the chunk is picked again -- all on the
    // second stream, no host round trip.  (Single calls redo such a chunk from the host, below.)
    const bool device_redo = overlap && sparse_ok && !needle_is_segmented(h, o) && o.device_redo != 0;
    if (device_redo) {   // sized once for the haystack with the most block pairs: no pick of the batch waits for an allocation
        for (int set = 0; set < 2; ++set)
            if ((rc = c->redo_pairs[set].ensure(sizeof(int) * (size_t)std::max<long long>(need.npairs, 1)))) return rc;
    }
    SegHeader* h_hdr = static_cast<SegHeader*>(c->hdr.p);
    auto chunk_events = [&](size_t k, int stage) {
        if (hooks.chunk_fn)
            for (int i = 0; i < n_chunks[k]; ++i)
                hooks.chunk_fn(hooks.chunk_user, G(k), (part ? part->chunk_base : 0) + (size_t)i,
                               part ? part->chunk_total : (size_t)n_chunks[k], stage);
    };
    size_t seq = 0;
    // Which path a failed chunk takes -- redone on the device, or from the host after the call -- depends on
    // when the first failure flag becomes visible to this loop: a race between host and GPU that no test can
    // steer.  The results are identical either way; "debug_redo_arm_at" pins the switch-over to a haystack
    // index (0: armed from the start, -1: never) so that both paths and the switch are tested deterministically.
    const bool arm_forced = o.debug_redo_arm_at >= -1;
    bool redo_armed = device_redo && (arm_forced ? o.debug_redo_arm_at == 0 : h->redo_armed_left[sm] > 0);
    QueueingScope queueing(o.debug_no_realloc != 0);
    for (size_t k = 0; k < n_hay; ++k) {
        const int ns = seg_off[k + 1] - seg_off[k];
        if (n_chunks[k] == 0) continue;
        if (device_redo && arm_forced) redo_armed = o.debug_redo_arm_at >= 0 && (long long)k >= o.debug_redo_arm_at;
        else if (device_redo && !redo_armed && (k & 3) == 0) {
            // (the flags of the haystacks queued so far: written by their picks, whenever those have run)
            const volatile unsigned char* f = h_fail;
            for (int i = 0; i < seg_off[k] && !redo_armed; ++i) redo_armed = f[i] != 0;
        }
        if (hooks.fn) hooks.fn(hooks.user, G(k), 0, (size_t)n_chunks[k]);
        chunk_events(k, 0);
        if (ns == 0) continue;
        const long long out_count = (long long)(lens[k] - s + 1);
        const int set = overlap ? (int)(seq & 1) : 0;
        float* d_scores = (float*)(set ? c->scores_b.p : c->scores.p);
        scan.set = set;
        // this set's work matrix, scores and summaries are overwritten: the pick (and redo) that last used them must be done
        // (On the host: this thread runs far ahead of the GPU -- it queues a haystack in 35 us, the GPU takes 700 -- so
        // waiting here for the pick of the haystack before the previous one leaves more than a haystack's work queued,
        // and the main stream is spared a barrier packet between K3 and the next K1: that boundary measured 6.5 us
        // instead of 11 - 27, profiles/r04/event_gaps.txt.  Option host_pick_wait = 0: the stream waits.)
        if (overlap && seq >= 2) {
            if (o.host_pick_wait) AM_HIP(hipEventSynchronize(c->ev_pick[set]));
            else AM_HIP(hipStreamWaitEvent(c->stream, c->ev_pick[set], 0));
        }
        scan.before_k3 = nullptr;
        scan.work_by_set = overlap;
        int* d_redo = nullptr;
        if (redo_armed) {
            Geometry g{};
            if ((rc = plan_geometry(s, out_count, o, &g))) return rc;
            AM_HIP(hipMemsetAsync(c->redo_pairs[set].p, 0, sizeof(int) * (size_t)g.npairs, c->stream));
            d_redo = static_cast<int*>(c->redo_pairs[set].p);
        }
        // (i16 frames are always finite -- but a half-precision transform can overflow on them)
        scan.bad = ((src_kind == 0 || o.half) && std::isfinite(factor)) ? &h_bad[k] : nullptr;
        if (pre) {
            // the pairs that were computed while the samples arrived are in the stream's own buffers:
            // only the rest is launched now, into the same buffers
            Geometry g{};
            if ((rc = plan_geometry(s, out_count, o, &g))) return rc;
            d_scores = pre->scores;
            scan.ext_stats32 = pre->stats32; scan.ext_side = pre->side;
            scan.side_nblocks = pre->layout_nblocks;
            scan.range_a = std::min(pre->pairs_done, g.npairs) * 2 * g.hop;
            scan.range_b = out_count;
            scan.skip_launch = scan.range_a >= out_count;
        }
        scan.tail_by_caller = false;
        if (batch_tails && tail_batchable(tails[k])) {
            if (tail_slot[k] < 0) {   // the next batch: this haystack and the following ones with such a tail
                std::vector<size_t> members;
                for (size_t k2 = k; k2 < n_hay && members.size() < (size_t)kMaxTailBatch; ++k2)
                    if (tail_batchable(tails[k2]) && tail_slot[k2] < 0) {
                        tail_slot[k2] = (tail_batches & 1) * kMaxTailBatch + (int)members.size();
                        members.push_back(k2);
                    }
                if ((rc = launch_tail_batch(h, o, tails, members, tail_batches & 1, tslots, d_hays, lens, factor, src_kind))) return rc;
                ++tail_batches;
            }
            scan.tail_by_caller = true;
        }
        if ((rc = run_correlation(h, o, d_hays[k], (long long)lens[k], 0, d_scores, out_count, factor,
                                  &scan, src_kind))) return rc;
        if (overlap) {
            AM_HIP(hipEventRecord(c->ev_k3[set], c->stream));
            AM_HIP(hipStreamWaitEvent(c->stream2, c->ev_k3[set], 0));
        }
        if (scan.tail_by_caller && scan.fused) {
            // (behind the main pass in stream order, hence behind the batch that filled the slot; in front of the pick)
            const TailPlan& t = tails[k];
            const size_t tiles = (size_t)1 << (scan.sparse.log_n2 - kColsLog), words = tiles << (scan.sparse.log_n1 - 6);
            const size_t blk = (size_t)(t.T / scan.sparse.hop);
            ProfScope ps(c, KN_OTHER, c->stream2);
            AM_HIP(launch_tail_commit(c->stream2, static_cast<const float*>(c->tail_scores.p) + (size_t)tail_slot[k] * tslots.scores, d_scores + t.T,
                                      out_count - t.T, static_cast<const float2*>(c->tail_stats.p) + (size_t)tail_slot[k] * tslots.stats,
                                      const_cast<float2*>(scan.sparse.stats32) + t.T / 32,
                                      scan.sparse.wbits ? const_cast<unsigned long long*>(scan.sparse.wbits) + blk * words : nullptr, (long long)words,
                                      scan.sparse.tile_theta ? const_cast<float*>(scan.sparse.tile_theta) + blk * tiles : nullptr, (int)tiles));
        }
        if (scan.fused && scan.sparse.wbits) {
            scan.sparse.fail_flags = h_fail + seg_off[k];
            scan.sparse.redo_pairs = (redo_armed && scan.redo_ok) ? d_redo : nullptr;
        }
        if ((rc = launch_pick(c, d_scores, out_count, seg_off[k], ns, p->min_prominence,
                              (long long)p->min_distance, &scan, seg_off[k], arena, pol, overlap ? c->stream2 : c->stream))) return rc;
        if (scan.fused && scan.sparse.redo_pairs) {
            ScanCfg cfg = scan.redo_cfg;
            cfg.margin = -1.0f;
            cfg.only_pairs = d_redo;
            { ProfScope ps(c, KN_OTHER, c->stream2);   // (not under "k3_cols_inv": an all-but-empty launch that queues behind the next haystack's kernels)
              AM_HIP(launch_k3(c->stream2, scan.redo_job, scan.redo_npairs, scan.redo_work, scan.redo_pl, scan.redo_scale, cfg, scan.redo_half)); }
            ScanRequest again = scan;
            again.sparse.redo_pairs = nullptr; again.sparse.fail_flags = nullptr; again.bad = nullptr;
            if ((rc = launch_pick(c, d_scores, out_count, seg_off[k], ns, p->min_prominence, (long long)p->min_distance,
                                  &again, seg_off[k], arena, pol, c->stream2, true))) return rc;
        }
        if (overlap) AM_HIP(hipEventRecord(c->ev_pick[set], c->stream2));
        ++seq;
    }
    queueing.end();
    AM_HIP(hipStreamSynchronize(c->stream));   // the headers are in host memory once the peak kernels have finished
    if (overlap) AM_HIP(hipStreamSynchronize(c->stream2));
    scan.set = 0;
    scan.before_k3 = nullptr;
    scan.bad = nullptr;
    scan.ext_stats32 = nullptr; scan.ext_side = nullptr; scan.side_nblocks = 0; scan.skip_launch = false;   // (the single-chunk passes below work in the context's own buffers)
    scan.tail_by_caller = false;   // (... and compute a tail they need themselves)
    scan.range_a = 0; scan.range_b = 0;
    int worst = AM_OK;
    std::vector<am_peak> all;
    std::vector<size_t> retry_f32;
    const int spare_hdr = (int)nsegs;
    for (size_t k = 0; k < n_hay; ++k) {
        const int s0 = seg_off[k], s1 = seg_off[k + 1];
        if (n_chunks[k] == 0) continue;
        const long long out_count = (long long)(lens[k] - s + 1);
        // Non-finite scores out of a half-precision pipeline: most likely an overflow of f16's range in the
        // row transform (a strong component that needle and haystack share, e.g. a DC offset or a steady
        // tone, concentrates in a few bins).  The haystack is matched again in f32, after every other
        // result of this call has been collected (the pass reuses the call's result area).
        if (h_bad[k] && o.half) { retry_f32.push_back(k); continue; }
        if (!my && !h_bad[k] && s1 > s0) {   // (a haystack with non-finite scores teaches the threshold nothing)
            std::vector<float> mins;
            int failed = 0;
            for (int i = s0; i < s1; ++i) { mins.push_back(h_hdr[i].seg_min); failed += h_fail[i] != 0; }
            std::sort(mins.begin(), mins.end());
            // many failed certificates: a score array that drifts (chunk minima in other block pairs than the tiles'
            // scores) -- the lowest minimum for a good while; a few, redone on the device: the background level
            if (failed * 8 > s1 - s0) h->conservative_left[sm] = 64;
            const bool robust = device_redo && h->conservative_left[sm] == 0;
            h->remember_min(sm, robust ? mins[mins.size() / 2] : mins.front());
            if (h->conservative_left[sm] > 0) --h->conservative_left[sm];
            if (failed) h->redo_armed_left[sm] = 64;
            else if (h->redo_armed_left[sm] > 0) --h->redo_armed_left[sm];
        }
        all.clear();
        // Non-finite samples (NaN, +-inf; f32 sources only).  The reference transforms every window
        // on its own (audio_matcher.rs:114-122): a window that holds such a sample gets NaN scores
        // throughout and yields no peak, every other window is untouched.  Here the sample has
        // poisoned the whole pair of overlap-save blocks around it, which reaches into neighbouring
        // chunks.  So, when a score kernel has reported a non-finite score for this haystack: find
        // the block pairs and the windows that hold such samples; a window that holds one yields
        // no peak; a clean window whose scores came from a poisoned pair is correlated again on its
        // own samples (as the reference does it) and picked from that.
        std::vector<char> drop, again;
        if (h_bad[k]) {
            if ((rc = classify_nonfinite(h, o, (const float*)d_hays[k], lens[k], out_count, segs, s0, s1, &drop, &again, pre == nullptr))) return rc;
        }
        // collect in window order (audio_matcher.rs:132-133)
        for (int i = s0; i < s1; ++i) {
            if (!drop.empty() && drop[i - s0]) continue;
            if (!again.empty() && again[i - s0]) {
                const Segment sg = segs[i];
                const long long cnt = sg.b - sg.a;
                ScanRequest one{};
                one.margin = -1.0f;
                PeakArena own{};
                if ((rc = c->spill.ensure(sizeof(am_peak) * AM_MAX_PEAKS_PER_CHUNK))) return rc;
                AM_HIP(hipMemsetAsync(c->arena_cur.p, 0, sizeof(unsigned), c->stream));
                own.base = static_cast<am_peak*>(c->spill.p); own.cursor = static_cast<unsigned*>(c->arena_cur.p);
                own.cap = AM_MAX_PEAKS_PER_CHUNK;
                if ((rc = run_correlation(h, o, advance_src(d_hays[k], (size_t)sg.a), (long long)widths[i], 0, (float*)c->scores.p, cnt,
                                          factor, &one, src_kind))) return rc;
                if ((rc = launch_pick(c, (const float*)c->scores.p, cnt, local_seg, 1, p->min_prominence,
                                      (long long)p->min_distance, &one, spare_hdr, own, pol))) return rc;
                AM_HIP(hipStreamSynchronize(c->stream));
                const SegHeader& hd = h_hdr[spare_hdr];
                const size_t old = all.size();
                if (hd.overflow & 1) {
                    if ((rc = pick_chunk_big(c, (const float*)c->scores.p, cnt, local_seg, Segment{0, cnt}, p->min_prominence,
                                             (long long)p->min_distance, &one, hd.seg_min, all, pol))) return rc;
                } else append_header_peaks(hd, own, all);
                for (size_t j = old; j < all.size(); ++j) { all[j].start += (uint64_t)sg.a; all[j].end += (uint64_t)sg.a; }
                continue;
            }
            if (!(h_hdr[i].overflow & 7)) { append_header_peaks(h_hdr[i], arena, all); continue; }
            // Rare: a write threshold was too high for this chunk (its minimum lies more than half a
            // prominence below the minimum some K3 tile sampled), its list found no room in the spill
            // arena, or more than AM_MAX_PEAKS_PER_CHUNK peaks passed the prominence filter (the
            // score buffers have moved on to later haystacks by now).  Redo the blocks that produce this
            // chunk's scores with every run written, in place in set 0 (same block layout, hence
            // bit-identical scores), and pick the chunk again with a spill arena of its own.
            ScanRequest full = scan;
            full.margin = -1.0f;
            full.range_a = segs[i].a; full.range_b = segs[i].b;
            PeakArena own{};
            if ((rc = c->spill.ensure(sizeof(am_peak) * AM_MAX_PEAKS_PER_CHUNK))) return rc;
            AM_HIP(hipMemsetAsync(c->arena_cur.p, 0, sizeof(unsigned), c->stream));
            own.base = static_cast<am_peak*>(c->spill.p); own.cursor = static_cast<unsigned*>(c->arena_cur.p);
            own.cap = AM_MAX_PEAKS_PER_CHUNK;
            if ((rc = run_correlation(h, o, d_hays[k], (long long)lens[k], 0, (float*)c->scores.p, out_count, factor,
                                      &full, src_kind))) return rc;
            if ((rc = launch_pick(c, (const float*)c->scores.p, out_count, i, 1, p->min_prominence,
                                  (long long)p->min_distance, &full, spare_hdr, own, pol))) return rc;
            AM_HIP(hipStreamSynchronize(c->stream));
            const SegHeader& hd = h_hdr[spare_hdr];
            if (hd.overflow & 1) {
                if ((rc = pick_chunk_big(c, (const float*)c->scores.p, out_count, i, segs[i], p->min_prominence,
                                         (long long)p->min_distance, &full, hd.seg_min, all, pol))) return rc;
            } else append_header_peaks(hd, own, all);
        }
        // second pass (MyConvolve scaling only): the shorter windows at the end of the haystack
        for (int i = tail_off[k]; i < tail_off[k + 1]; ++i) {
            const Segment sg = tail_segs[i];
            const long long cnt = sg.b - sg.a;
            ScanRequest one{};
            one.margin = -1.0f;
            one.seg_c = 0; one.seg_d = 0;
            PeakArena own{};
            if ((rc = c->spill.ensure(sizeof(am_peak) * AM_MAX_PEAKS_PER_CHUNK))) return rc;
            AM_HIP(hipMemsetAsync(c->arena_cur.p, 0, sizeof(unsigned), c->stream));
            own.base = static_cast<am_peak*>(c->spill.p); own.cursor = static_cast<unsigned*>(c->arena_cur.p);
            own.cap = AM_MAX_PEAKS_PER_CHUNK;
            if ((rc = c->scores.ensure((size_t)cnt * sizeof(float)))) return rc;
            if ((rc = run_correlation(h, o, advance_src(d_hays[k], (size_t)sg.a), (long long)tail_w[i], 0, (float*)c->scores.p, cnt,
                                      scale_factor(h, p->scale, tail_w[i]), &one, src_kind))) return rc;
            if ((rc = launch_pick(c, (const float*)c->scores.p, cnt, (int)nsegs + i, 1, p->min_prominence,
                                  (long long)p->min_distance, &one, spare_hdr, own, pol))) return rc;
            AM_HIP(hipStreamSynchronize(c->stream));
            const SegHeader& hd = h_hdr[spare_hdr];
            const size_t old = all.size();
            if (hd.overflow & 1) {
                if ((rc = pick_chunk_big(c, (const float*)c->scores.p, cnt, (int)nsegs + i, Segment{0, cnt}, p->min_prominence,
                                         (long long)p->min_distance, &one, hd.seg_min, all, pol))) return rc;
            } else append_header_peaks(hd, own, all);
            for (size_t j = old; j < all.size(); ++j) { all[j].start += (uint64_t)sg.a; all[j].end += (uint64_t)sg.a; }   // audio_matcher.rs:126
        }
        if (part) {   // unmerged, in window order (audio_matcher.rs:132-133), at their positions in the whole haystack
            for (am_peak& q : all) { q.start += part->first_sample; q.end += part->first_sample; }
            part->raw->insert(part->raw->end(), all.begin(), all.end());
            n_out[G(k)] = all.size();
            rc = AM_OK;
        } else rc = merge_peaks(all, p, o.surrounding_from != 0, out ? out + G(k) * cap_per_hay : nullptr, cap_per_hay, &n_out[G(k)]);
        chunk_events(k, 1);
        if (hooks.fn) hooks.fn(hooks.user, G(k), 1, (size_t)n_chunks[k]);
        if (rc == AM_ERR_CAPACITY) worst = rc;
        else if (rc) return rc;
    }
    for (size_t k : retry_f32) {
        const long long keep = h->opt_half;
        h->opt_half = 0;
        rc = match_many(h, &d_hays[k], &lens[k], 1, p, out ? out + G(k) * cap_per_hay : nullptr, cap_per_hay, &n_out[G(k)], src_kind,
                        0, 1, false, nullptr, part);
        h->opt_half = keep;
        chunk_events(k, 1);
        if (hooks.fn) hooks.fn(hooks.user, G(k), 1, (size_t)n_chunks[k]);
        if (rc == AM_ERR_CAPACITY) worst = rc;
        else if (rc) return rc;
    }
    return worst;
}


// BASELINE config 4: several needles against a batch of resident haystacks = the per-file loop of
// matcher::run (matcher/mod.rs:42-87) around N snippets.  Per haystack the forward column pass (K1)
// runs once; needles are then taken in groups that share the forward row transforms of K2
// (k2_rows_r16_group), each needle with its own inverse rows, K3 (fused scan) and peak pick.  The
// pick of (haystack, needle) runs on the second stream beside the next needle's K3 / the next
// haystack's K1 and K2; the score-side buffers alternate between two sets, as in match_many.
// Needles must share one length so that they share the block layout.
// Result slot of (haystack k of the caller's batch, needle j): G(k) * nn + j.
static int match_multi_many(am_needle* const* needles, size_t nn, const void* const* d_hays, const size_t* lens, size_t n_hay,
                            int src_kind, const am_match_params* p, am_peak* out, size_t cap_per_pair, size_t* n_out,
                            size_t index_base = 0, size_t index_stride = 1) {
    am_needle* h0 = needles[0];
    Ctx* c = h0->ctx;
    const Opts o = snapshot_opts(h0);
    const PeakPolicy pol = o.peak_policy();
    const Hooks hooks = snapshot_hooks();
    auto G = [&](size_t k) { return index_base + k * index_stride; };
    const size_t s = h0->n;
    for (size_t j = 0; j < nn; ++j) {
        if (!needles[j] || needles[j]->ctx != c) return fail(AM_ERR_INVALID_ARG, "needles must live on one device");
        if (needles[j]->n != s) return fail(AM_ERR_INVALID_ARG, "am_match_multi: needles must have equal length");
    }
    if (p->chunk == 0) return fail(AM_ERR_INVALID_ARG, "chunk must be > 0");
    if (p->scale != AM_SCALE_NONE && p->scale != AM_SCALE_LIB)
        return fail(AM_ERR_INVALID_ARG, "am_match_multi supports AM_SCALE_NONE and AM_SCALE_LIB");
    for (size_t k = 0; k < n_hay; ++k)
        for (size_t j = 0; j < nn; ++j) n_out[G(k) * nn + j] = 0;
    if (needle_is_segmented(h0, o)) {
        // partitioned needles (longer than kSegmentFrom samples) share nothing here: pair by pair
        int worst = AM_OK;
        for (size_t k = 0; k < n_hay; ++k)
            for (size_t j = 0; j < nn; ++j) {
                const size_t slot = G(k) * nn + j;
                const int rc = match_many(needles[j], &d_hays[k], &lens[k], 1, p, out ? out + slot * cap_per_pair : nullptr, cap_per_pair,
                                          &n_out[slot], src_kind, 0, 1, false);
                if (rc == AM_ERR_CAPACITY) worst = rc;
                else if (rc) return rc;
            }
        return worst;
    }
    const int sm = p->scale == AM_SCALE_LIB ? 1 : 0;
    // the chunk lists of every haystack, back to back, and each haystack's block layout
    std::vector<Segment> segs;
    std::vector<int> seg_off(n_hay + 1, 0);
    std::vector<Geometry> geo(n_hay);
    std::vector<TailPlan> tails(n_hay);   // the odd last block on the 2^21 plan (TailPlan), for haystacks whose needle groups all take the grouped K3
    size_t max_scores = 0, max_segs = 1, max_work = 0, max_matrix = 0, max_wflags = 0, max_tail = 0;
    int rc;
    for (size_t k = 0; k < n_hay; ++k) {
        seg_off[k] = (int)segs.size();
        if (!d_hays[k] || lens[k] < s) continue;
        make_segments(lens[k], s, p, o.tail_window != 0, segs);
        const size_t ns = segs.size() - (size_t)seg_off[k];
        if (ns == 0) continue;
        const long long out_count = (long long)(lens[k] - s + 1);
        if ((rc = plan_geometry(s, out_count, o, &geo[k]))) return rc;
        const size_t matrix = (size_t)geo[k].npairs * (size_t)geo[k].N;
        max_scores = std::max(max_scores, (size_t)out_count);
        max_segs = std::max(max_segs, ns);
        max_work = std::max(max_work, matrix);
        max_matrix = std::max(max_matrix, matrix);
        { const Plan* plk = nullptr; if ((rc = get_plan(c, geo[k].logN, &plk))) return rc; max_wflags = std::max(max_wflags, sparse_bytes(geo[k].nblocks, plk->dev)); }
        tail_plan(s, out_count, o, geo[k], &tails[k]);
    }
    seg_off[n_hay] = (int)segs.size();
    const size_t nsegs = segs.size();
    if (nsegs == 0) return AM_OK;
    if (max_segs > (size_t)1 << 18 || nsegs * nn > (size_t)1 << 24)
        return fail(AM_ERR_INVALID_ARG, "chunk size too small for this batch (too many chunks)");
    // every needle's spectrum for every plan in use, before the work matrix is filled (building one uses it)
    std::map<int, std::vector<const float2*>> hcs;
    for (size_t k = 0; k < n_hay; ++k) {
        if (seg_off[k + 1] == seg_off[k] || hcs.count(geo[k].logN)) continue;
        const Plan* pl = nullptr;
        if ((rc = get_plan(c, geo[k].logN, &pl))) return rc;
        std::vector<const float2*>& v = hcs[geo[k].logN];
        v.resize(nn);
        for (size_t j = 0; j < nn; ++j) {
            if ((rc = needle_spectrum(needles[j], pl, &v[j]))) return rc;
            const HalfScale hs = half_scale(needles[j], o, pl->dev);
            if (hs.level == 2 && (rc = needle_spectrum16(needles[j], pl, hs.hscale, &v[j]))) return rc;
        }
    }
    const size_t group_opt = (size_t)std::min<long long>(std::max<long long>(1, o.needle_group), kMaxNeedleGroup);
    // The tail needs every needle group of the haystack on the grouped-K3 path (the other paths keep the full layout):
    // f32, groups of at least two needles each, the 512-row plan with a 256-row tail.
    {
        const bool groups_ok = o.k3_group && group_opt > 1 && nn > 1 && !o.half && (nn % group_opt) != 1 && c->stream_tail != nullptr;
        for (size_t k = 0; k < n_hay; ++k) {
            if (!tails[k].on) continue;
            const Plan* plk = nullptr;
            if (seg_off[k + 1] == seg_off[k] || !groups_ok || !tail_batchable(tails[k]) || get_plan(c, geo[k].logN, &plk) || !plan_is_c512(plk->dev)) {
                tails[k].on = false;
                continue;
            }
            max_tail = std::max(max_tail, (size_t)tails[k].g.N);
            if (!hcs.count(tails[k].g.logN)) {   // the needles' spectra on the tail's plan
                const Plan* plt = nullptr;
                if ((rc = get_plan(c, tails[k].g.logN, &plt))) return rc;
                std::vector<const float2*>& v = hcs[tails[k].g.logN];
                v.resize(nn);
                for (size_t j = 0; j < nn; ++j)
                    if ((rc = needle_spectrum(needles[j], plt, &v[j]))) return rc;
            }
        }
        if (max_tail) {
            if ((rc = c->work_tail.ensure(std::max(c->work_tail.cap, max_tail * sizeof(float2))))) return rc;
            if ((rc = c->work_tail2.ensure(std::min(group_opt, nn) * max_tail * sizeof(float2)))) return rc;
        }
    }
    size_t n_pairs_active = 0;
    for (size_t k = 0; k < n_hay; ++k) n_pairs_active += seg_off[k + 1] > seg_off[k] ? nn : 0;
    const bool overlap = o.batch_overlap && n_pairs_active > 1 && c->stream2 &&
                         c->ev_k3[0] && c->ev_k3[1] && c->ev_pick[0] && c->ev_pick[1];
    if ((rc = c->work.ensure(max_work * sizeof(float2)))) return rc;
    if ((rc = c->work2.ensure(std::min(group_opt, nn) * max_matrix * sizeof(float2)))) return rc;
    for (int set = 0; set < (overlap ? 2 : 1); ++set) {
        if ((rc = (set ? c->scores_b : c->scores).ensure(max_scores * sizeof(float)))) return rc;
        if ((rc = (set ? c->peaks_b : c->peaks).ensure(sizeof(am_peak) * max_segs * AM_MAX_PEAKS_PER_CHUNK))) return rc;
        if ((rc = (set ? c->stats_b : c->stats).ensure((max_scores + kTile - 1) / kTile * sizeof(float2)))) return rc;
        if ((rc = (set ? c->stats32_b : c->stats32).ensure((max_scores + 31) / 32 * sizeof(float2)))) return rc;
        if ((rc = (set ? c->wflags_b : c->wflags).ensure(max_wflags))) return rc;
    }
    if ((rc = c->wide_ctl.ensure(max_segs * 24))) return rc;
    if ((rc = c->wide_list.ensure(max_segs * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
    if ((rc = c->wide_tiles.ensure(max_segs * kWideTileList * sizeof(int)))) return rc;
    // one K3 launch per needle group: every needle of a group (two groups in flight) has its own score-side buffers
    const size_t k3_group = (o.k3_group && group_opt > 1 && nn > 1 && !o.half && max_wflags > 0) ? std::min(group_opt, nn) : 0;
    for (size_t i = 0; i < k3_group * (overlap ? 2 : 1); ++i) {
        const size_t slot = i < k3_group ? i : kMaxNeedleGroup + (i - k3_group);
        if ((rc = c->grp_scores[slot].ensure(max_scores * sizeof(float)))) return rc;
        if ((rc = c->grp_stats32[slot].ensure((max_scores + 31) / 32 * sizeof(float2)))) return rc;
        if ((rc = c->grp_wflags[slot].ensure(max_wflags))) return rc;
    }
    if (k3_group && o.pick_group) {   // ... and the scratch of the group's picks, which run as one set of launches
        const size_t total = k3_group * max_segs;
        for (size_t z = 0; z < k3_group; ++z)
            if ((rc = c->grp_stats[z].ensure((max_scores + kTile - 1) / kTile * sizeof(float2)))) return rc;
        if ((rc = c->wide_ctl.ensure(total * 24))) return rc;
        if ((rc = c->wide_list.ensure(total * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
        if ((rc = c->wide_tiles.ensure(total * kWideTileList * sizeof(int)))) return rc;
        if ((rc = c->peaks.ensure(total * AM_MAX_PEAKS_PER_CHUNK * sizeof(am_peak)))) return rc;
    }
    PeakArena arena{};
    if ((rc = prepare_results(c, nsegs * nn, nsegs * nn * 8 + 4096, &arena))) return rc;
    if ((rc = upload_segments(c, segs))) return rc;
    if ((rc = c->badflag.ensure(sizeof(int) * n_hay))) return rc;
    int* h_bad = static_cast<int*>(c->badflag.p);
    memset(h_bad, 0, sizeof(int) * n_hay);
    SegHeader* h_hdr = static_cast<SegHeader*>(c->hdr.p);
    // result headers of (haystack k, needle j): nsegs entries per needle, the haystack's slice inside
    auto hdr_of = [&](size_t k, size_t j) { return (int)(j * nsegs) + seg_off[k]; };
    size_t seq = 0;
    QueueingScope queueing(o.debug_no_realloc != 0);
    for (size_t k = 0; k < n_hay; ++k) {
        const int ns = seg_off[k + 1] - seg_off[k];
        if (ns == 0) continue;
        if (hooks.fn) hooks.fn(hooks.user, G(k), 0, (size_t)ns);
        const Geometry& g = geo[k];
        const Plan* pl = nullptr;
        if ((rc = get_plan(c, g.logN, &pl))) return rc;
        const std::vector<const float2*>& hc = hcs[g.logN];
        const long long out_count = (long long)(lens[k] - s + 1);
        const int half = (o.half && (plan_is_r16(pl->dev) || plan_is_c512(pl->dev))) ? (o.half >= 2 ? 2 : 1) : 0;
        const size_t group = (!half && plan_k2_has_group(pl->dev)) ? group_opt : 1;
        const size_t matrix = (size_t)g.npairs * (size_t)g.N;
        const bool fused = plan_has_scan(pl->dev) && (g.hop % kTile) == 0;
        // The odd last block (TailPlan): the main pass -- K1 here, every group's K2 and K3 below -- stops at the even
        // block boundary, the scores behind it come from one pair of the 2^21 plan: K1 once, then per needle group one
        // row-kernel launch and one K3 launch (every run written) behind the group's own, and the main layout's
        // ballots / thresholds of that block preset for the group's needles.
        const TailPlan& tail = tails[k];
        const int main_pairs = (int)(tail.on ? g.npairs - 1 : g.npairs);
        const Plan* plt = nullptr;
        Job job{}, job_t{};
        job.src = d_hays[k]; job.src_len = (long long)lens[k]; job.lead = 0; job.src_kind = src_kind;
        job.out_count = tail.on ? tail.T : out_count; job.hop = (int)g.hop; job.nblocks = (int)(tail.on ? g.nblocks - 1 : g.nblocks); job.first_pair = 0;
        { ProfScope ps(c, KN_K1); AM_HIP(launch_k1(c->stream, job, main_pairs, (float2*)c->work.p, pl->dev, half)); }
        if (tail.on) {
            if ((rc = get_plan(c, tail.g.logN, &plt))) return rc;
            job_t.src = static_cast<const char*>(d_hays[k]) + 4 * (size_t)tail.T; job_t.src_len = (long long)lens[k] - tail.T; job_t.lead = 0;
            job_t.src_kind = src_kind; job_t.out_count = out_count - tail.T; job_t.hop = (int)tail.g.hop; job_t.nblocks = (int)tail.g.nblocks;
            job_t.first_pair = 0;
            ProfScope ps(c, KN_OTHER);
            AM_HIP(launch_k1(c->stream, job_t, 1, (float2*)c->work_tail.p, plt->dev, 0));
        }
        for (size_t j = 0; j < nn; ++j) {
            am_needle* h = needles[j];
            const size_t in_group = j % group;
            const float2* inv_rows = (const float2*)c->work2.p + in_group * matrix;   // this needle's inverse rows
            const size_t gn = std::min(group, nn - (j - in_group));
            if (group > 1 && in_group == 0) {
                K2Group grp{};
                grp.n = (int)gn;
                for (int q = 0; q < grp.n; ++q) { grp.hc[q] = hc[j + q]; grp.dst[q] = (float2*)c->work2.p + (size_t)q * matrix; }
                { ProfScope ps(c, KN_K2); AM_HIP(launch_k2_group(c->stream, main_pairs, (const float2*)c->work.p, grp, pl->dev)); }
                if (tail.on) {
                    K2Group gt{};
                    gt.n = (int)gn;
                    const std::vector<const float2*>& hct = hcs[tail.g.logN];
                    for (int q = 0; q < gt.n; ++q) { gt.hc[q] = hct[j + q]; gt.dst[q] = (float2*)c->work_tail2.p + (size_t)q * (size_t)tail.g.N; }
                    ProfScope ps(c, KN_OTHER);
                    AM_HIP(launch_k2_group(c->stream, 1, (const float2*)c->work_tail.p, gt, plt->dev));
                }
            }
            // The K3s of the group as one launch (needle index on blockIdx.y), the group's picks queued behind it.
            const bool grouped_k3 = k3_group && group > 1 && gn > 1 && fused && !half && plan_k3_has_group(pl->dev);
            if (grouped_k3 && in_group != 0) continue;   // (handled with the group's first needle)
            const int set = overlap ? (int)(seq & 1) : 0;
            const float margin = (!o.dense && p->min_prominence > 0.f) ? 0.5f * p->min_prominence : -1.0f;
            if (grouped_k3) {
                K3Group kg{};
                kg.n = (int)gn;
                ScanRequest scans[kMaxNeedleGroup];
                ScanCfg common{};
                for (size_t q = 0; q < gn; ++q) {
                    am_needle* hq = needles[j + q];
                    const size_t slot = (size_t)set * kMaxNeedleGroup + q;
                    ScanRequest& sc = scans[q];
                    sc = ScanRequest{};
                    sc.set = 0;          // (the picks of a call run one after the other: they share the pick's own scratch)
                    sc.margin = margin; sc.hist_min = hq->hist_min(sm);
                    sc.seg_c = (long long)p->chunk; sc.seg_d = (long long)(p->chunk + p->overlap) - (long long)s;
                    sc.bad = (src_kind == 0 || o.half) ? &h_bad[k] : nullptr;
                    sc.fused = true;
                    ScanCfg cfg{};
                    fill_scan_cfg(&cfg, c->grp_stats32[slot].p, c->grp_wflags[slot].p, g.nblocks, pl->dev, margin, sc.hist_min, sc.seg_c, sc.seg_d);
                    sc.sparse = sparse_view(cfg, g.hop, pl->dev);
                    if (q == 0) common = cfg;
                    kg.work[q] = (const float2*)c->work2.p + q * matrix;
                    kg.dst[q] = (float*)c->grp_scores[slot].p;
                    kg.stats32[q] = cfg.stats32; kg.wbits[q] = cfg.wbits; kg.tile_theta[q] = cfg.tile_theta;
                    kg.hist_min[q] = cfg.hist_min;
                    kg.out_scale[q] = half_scale(hq, o, pl->dev).k3(scale_factor(hq, p->scale, 1));
                }
                // K3 overwrites this set's scores and summaries: the picks that last read them must be done
                if (overlap && seq >= 2) AM_HIP(hipStreamWaitEvent(c->stream, c->ev_pick[set], 0));
                { ProfScope ps(c, KN_K3); AM_HIP(launch_k3_group(c->stream, job, main_pairs, kg, pl->dev, common)); }
                if (tail.on) {
                    K3Group kt = kg;
                    for (size_t q = 0; q < gn; ++q) {
                        kt.work[q] = (const float2*)c->work_tail2.p + q * (size_t)tail.g.N;
                        kt.dst[q] = kg.dst[q] + tail.T; kt.stats32[q] = kg.stats32[q] + tail.T / 32;
                        kt.wbits[q] = nullptr; kt.tile_theta[q] = nullptr; kt.hist_min[q] = FLT_MAX;
                    }
                    ScanCfg dense{};
                    dense.stats32 = kt.stats32[0]; dense.margin = -1.0f; dense.hist_min = FLT_MAX;
                    ProfScope ps(c, KN_OTHER);
                    AM_HIP(launch_k3_group(c->stream, job_t, 1, kt, plt->dev, dense));
                    if (margin >= 0.0f)
                        AM_HIP(launch_tail_preset_group(c->stream, kg, (long long)(g.nblocks - 1), pl->dev.logN1, pl->dev.logN2));
                }
                if (overlap) {
                    AM_HIP(hipEventRecord(c->ev_k3[set], c->stream));
                    AM_HIP(hipStreamWaitEvent(c->stream2, c->ev_k3[set], 0));
                }
                if (o.pick_group) {
                    int hoff[kMaxNeedleGroup];
                    for (size_t q = 0; q < gn; ++q) hoff[q] = hdr_of(k, j + q);
                    if ((rc = launch_pick_group(c, kg, out_count, seg_off[k], ns, p->min_prominence, (long long)p->min_distance, scans[0].sparse,
                                                scans[0].bad, hoff, arena, pol, overlap ? c->stream2 : c->stream))) return rc;
                } else
                for (size_t q = 0; q < gn; ++q)
                    if ((rc = launch_pick(c, kg.dst[q], out_count, seg_off[k], ns, p->min_prominence, (long long)p->min_distance,
                                          &scans[q], hdr_of(k, j + q), arena, pol, overlap ? c->stream2 : c->stream))) return rc;
                if (overlap) AM_HIP(hipEventRecord(c->ev_pick[set], c->stream2));
                ++seq;
                continue;
            }
            float* d_scores = (float*)(set ? c->scores_b.p : c->scores.p);
            job.dst = d_scores;
            ScanRequest scan{};
            scan.set = set;
            scan.margin = margin;
            scan.hist_min = h->hist_min(sm);
            scan.seg_c = (long long)p->chunk;
            scan.seg_d = (long long)(p->chunk + p->overlap) - (long long)s;
            scan.bad = (src_kind == 0 || o.half) ? &h_bad[k] : nullptr;   // (i16 frames are always finite; an f16 transform can overflow)
            scan.fused = fused;
            scan.sparse = SparseScores{nullptr, nullptr, nullptr, (int)g.hop, pl->dev.logN2, pl->dev.logN1, 1.0 / (double)g.hop};
            ScanCfg cfg{};
            if (fused) {
                fill_scan_cfg(&cfg, set ? c->stats32_b.p : c->stats32.p, set ? c->wflags_b.p : c->wflags.p, g.nblocks, pl->dev, scan.margin,
                              scan.hist_min, scan.seg_c, scan.seg_d);
                scan.sparse = sparse_view(cfg, g.hop, pl->dev);
            }
            const float factor = scale_factor(h, p->scale, 1);
            const HalfScale hs = half_scale(h, o, pl->dev);
            if (group == 1) {
                ProfScope ps(c, KN_K2);
                AM_HIP(launch_k2(c->stream, (int)g.npairs, (float2*)c->work.p, hc[j], pl->dev, (float2*)c->work2.p, hs.level, hs.hscale, hs.pre));
            }
            // K3 overwrites this set's scores and summaries: the pick that last read them must be done
            if (overlap && seq >= 2) AM_HIP(hipStreamWaitEvent(c->stream, c->ev_pick[set], 0));
            { ProfScope ps(c, KN_K3); AM_HIP(launch_k3(c->stream, job, (int)g.npairs, inv_rows, pl->dev, hs.k3(factor), cfg, half)); }
            if (overlap) {
                AM_HIP(hipEventRecord(c->ev_k3[set], c->stream));
                AM_HIP(hipStreamWaitEvent(c->stream2, c->ev_k3[set], 0));
            }
            if ((rc = launch_pick(c, d_scores, out_count, seg_off[k], ns, p->min_prominence, (long long)p->min_distance,
                                  &scan, hdr_of(k, j), arena, pol, overlap ? c->stream2 : c->stream))) return rc;
            if (overlap) AM_HIP(hipEventRecord(c->ev_pick[set], c->stream2));
            ++seq;
        }
    }
    queueing.end();
    AM_HIP(hipStreamSynchronize(c->stream));
    if (overlap) AM_HIP(hipStreamSynchronize(c->stream2));
    int worst = AM_OK;
    std::vector<am_peak> all;
    std::vector<std::pair<size_t, size_t>> redo;
    for (size_t k = 0; k < n_hay; ++k) {
        const int ns = seg_off[k + 1] - seg_off[k];
        if (ns == 0) continue;
        for (size_t j = 0; j < nn; ++j) {
            const SegHeader* hd = h_hdr + hdr_of(k, j);
            const size_t slot = G(k) * nn + j;
            // Non-finite samples poison whole block pairs for every needle (see match_many): such a
            // haystack goes through the single-needle path, which gives every window the reference's
            // answer.  So does a pair with a failed certificate, a lost spill or more than
            // AM_MAX_PEAKS_PER_CHUNK peaks in a chunk.
            bool again = h_bad[k] != 0;
            float lowest = FLT_MAX;
            for (int i = 0; i < ns && !again; ++i) {
                if (hd[i].overflow & 7) again = true;
                lowest = std::min(lowest, hd[i].seg_min);
            }
            if (!again) needles[j]->remember_min(sm, lowest);
            if (again) { redo.emplace_back(k, j); continue; }
            all.clear();
            for (int i = 0; i < ns; ++i) append_header_peaks(hd[i], arena, all);
            rc = merge_peaks(all, p, o.surrounding_from != 0, out ? out + slot * cap_per_pair : nullptr, cap_per_pair, &n_out[slot]);
            if (rc == AM_ERR_CAPACITY) worst = rc;
            else if (rc) return rc;
        }
    }
    // the single-needle path reuses the result area: it runs after everything else has been collected
    for (const auto& kj : redo) {
        const size_t slot = G(kj.first) * nn + kj.second;
        n_out[slot] = 0;
        rc = match_many(needles[kj.second], &d_hays[kj.first], &lens[kj.first], 1, p, out ? out + slot * cap_per_pair : nullptr,
                        cap_per_pair, &n_out[slot], src_kind, 0, 1, false);
        if (rc == AM_ERR_CAPACITY) worst = rc;
        else if (rc) return rc;
    }
    if (hooks.fn)
        for (size_t k = 0; k < n_hay; ++k)
            if (seg_off[k + 1] > seg_off[k]) hooks.fn(hooks.user, G(k), 1, (size_t)(seg_off[k + 1] - seg_off[k]));
    return worst;
}

// find_peaks on one host score array (am_find_peaks)
static int find_peaks_host_array(Ctx* c, const float* d_scores, long long n, float min_prom, long long min_dist,
                                 std::vector<am_peak>& all) {
    int rc;
    const PeakPolicy pol = snapshot_opts(nullptr).peak_policy();
    Segment sg; sg.a = 0; sg.b = n;
    PeakArena arena{};
    if ((rc = prepare_results(c, 1, AM_MAX_PEAKS_PER_CHUNK, &arena))) return rc;
    if ((rc = c->peaks.ensure(sizeof(am_peak) * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    if ((rc = upload_segments(c, std::vector<Segment>(1, sg)))) return rc;
    if ((rc = launch_pick(c, d_scores, n, 0, 1, min_prom, min_dist, nullptr, 0, arena, pol))) return rc;
    AM_HIP(hipStreamSynchronize(c->stream));
    const SegHeader hd = *static_cast<const SegHeader*>(c->hdr.p);
    all.clear();
    if (hd.overflow) return pick_chunk_big(c, d_scores, n, 0, sg, min_prom, min_dist, nullptr, hd.seg_min, all, pol);
    append_header_peaks(hd, arena, all);
    return AM_OK;
}

static int check_needle(const am_needle* h) {
    if (!h || !h->ctx) return fail(AM_ERR_INVALID_ARG, "null needle handle");
    AM_HIP(hipSetDevice(h->ctx->device));
    return AM_OK;
}

static int create_needle_common(Ctx* c, float* d_needle, size_t n, am_needle** out) {
    am_needle* h = new am_needle();
    h->ctx = c; h->d_needle = d_needle; h->n = n;
    const int parts = sumsq_parts((long long)n);
    int rc = c->sum.ensure(sizeof(double) * (size_t)parts);
    if (rc) { (void)hipFree(d_needle); delete h; return rc; }
    hipError_t e = launch_sumsq(c->stream, d_needle, (long long)n, (double*)c->sum.p);
    std::vector<double> part((size_t)parts, 0.0);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = copy_on_stream(c, part.data(), c->sum.p, sizeof(double) * (size_t)parts, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { (void)hipFree(d_needle); delete h; return hip_fail(e, "needle energy"); }
    double ss = 0.0;
    for (double v : part) ss += v;
    h->inv_autocorr = (float)(1.0 / ss);   // audio_matcher.rs:321-329
    *out = h;
    return AM_OK;
}

}  // namespace am

using namespace am;

// ---------------------------------------------------------------------------
// The haystack batch over several devices (matcher/mod.rs:42-87 sharded).
struct am_pool {
    struct Slot {
        int device = -1;
        am_needle* needle = nullptr;           // needles[0]
        std::vector<am_needle*> needles;       // every needle of the pool, replicated on this device
        // two-slot HBM ring + copy stream of the host-buffer path
        void* ring[2] = {nullptr, nullptr};
        size_t ring_cap = 0;
        hipStream_t copy_stream = nullptr;
    };
    std::vector<Slot> slots;
    std::mutex mu;   // one batch at a time per pool
};


// ---------------------------------------------------------------------------
// Streaming ingest: calc_chunks consumes a lazy ExactSizeIterator (audio_matcher.rs:88-97; the
// windows are cut as the decoder yields frames, :104, mp3_reader.rs:13-41).  The stream object owns
// the haystack's device buffer and a set of score-side buffers; am_match_stream_push copies a block
// of samples on a copy stream and launches K1 / K2 / K3 for every block pair whose samples have
// arrived completely, so transfer (or decoding) and transforms overlap.
struct am_stream {
    am_needle* h = nullptr;
    int fmt = AM_FMT_F32_MONO;
    am_match_params p{};
    am::DevBuf hay, scores, stats32, side;
    size_t cap = 0, len = 0;          // elements (f32 samples or stereo frames, 4 bytes each); len = accepted so far
    size_t sent = 0;                  // elements whose host-to-device copy has been issued (len - sent sit in the staging ring)
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied = nullptr;
    // Two-slot staging ring in pinned host memory: a push of a decoder-sized piece (minimp3 yields 1152 frames,
    // mp3_reader.rs:28-37) is a host memcpy into the current slot and returns; a full slot goes to the device as one
    // asynchronous copy while the other slot fills.  Large pushes bypass the ring (one copy straight from the caller's
    // buffer, at link speed when that buffer is pinned: am_host_alloc / am_host_register).
    static constexpr size_t kStageElems = (size_t)1 << 20;    // 4 MB per slot
    static constexpr size_t kDirectElems = (size_t)1 << 18;   // pushes of 1 MB and more are copied directly
    am::HostBuf stage[2];
    hipEvent_t staged[2] = {nullptr, nullptr};                 // the slot's last copy has left it
    bool stage_busy[2] = {false, false};
    int cur = 0;
    size_t fill = 0;                  // elements in the current slot
    bool early = false;               // block pairs may be launched before the length is known
    long long pairs_done = 0;
    am::Geometry geo{};               // the provisional block layout (from the capacity)
    float margin = 0.f;               // the write-threshold margin the early pairs were launched with (< 0: every run written)
    bool failed = false;
};

namespace am {

// (re)computes the provisional layout for the stream's capacity and sizes its score-side buffers
static int stream_layout(am_stream* st) {
    am_needle* h = st->h;
    const Opts o = snapshot_opts(h);
    st->early = false;
    st->pairs_done = 0;
    if (st->cap < h->n || st->p.scale == AM_SCALE_MY || st->p.chunk == 0) return AM_OK;
    if (h->n <= (size_t)kDirectMaxNeedle && o.log_n == 0) return AM_OK;               // direct summation: no blocks
    if (o.log_n == 0 && (long long)h->n > kWidestFromSamples) return AM_OK;    // the plan depends on the final length / the needle is partitioned
    const long long out_cap = (long long)(st->cap - h->n + 1);
    int rc = plan_geometry(h->n, out_cap, o, &st->geo);
    if (rc) return rc;
    const Plan* pl = nullptr;
    if ((rc = get_plan(h->ctx, st->geo.logN, &pl))) return rc;
    if (!(plan_has_scan(pl->dev) && (st->geo.hop % kTile) == 0)) return AM_OK;           // (small generic plans: nothing to overlap)
    if ((rc = st->scores.ensure((size_t)out_cap * sizeof(float)))) return rc;
    if ((rc = st->stats32.ensure((size_t)((out_cap + 31) / 32) * sizeof(float2)))) return rc;
    if ((rc = st->side.ensure(sparse_bytes(st->geo.nblocks, pl->dev)))) return rc;
    st->early = true;
    return AM_OK;
}

}  // namespace am

// ===========================================================================
extern "C" {

int am_abi_version(void) { return AM_ABI_VERSION; }
const char* am_last_error_string(void) { return t_err.c_str(); }

int am_device_count(int* n) {
    if (!n) return fail(AM_ERR_INVALID_ARG, "null pointer");
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { *n = 0; return fail(AM_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = k;
    return AM_OK;
}

int am_needle_create(int device, const float* needle, size_t n, am_needle** out) {
    if (!needle || !out || n == 0) return fail(AM_ERR_INVALID_ARG, "needle must be non-empty");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    float* d = nullptr;
    AM_HIP(hipMalloc((void**)&d, n * sizeof(float)));
    hipError_t e = copy_on_stream(c, d, needle, n * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "copy_on_stream(c, needle)"); }
    return create_needle_common(c, d, n, out);
}

int am_needle_create_device(int device, const float* d_needle, size_t n, am_needle** out) {
    if (!d_needle || !out || n == 0) return fail(AM_ERR_INVALID_ARG, "needle must be non-empty");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    float* d = nullptr;
    AM_HIP(hipMalloc((void**)&d, n * sizeof(float)));
    hipError_t e = copy_on_stream(c, d, d_needle, n * sizeof(float), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "copy_on_stream(c, needle d2d)"); }
    return create_needle_common(c, d, n, out);
}

void am_needle_destroy(am_needle* h) {
    if (!h) return;
    if (h->ctx) {
        std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
        (void)hipSetDevice(h->ctx->device);
        (void)hipStreamSynchronize(h->ctx->stream);
        for (am_needle* sub : h->segments) {
            for (auto& kv : sub->spectra) (void)hipFree(kv.second);
            for (auto& kv : sub->spectra16) (void)hipFree(kv.second);
            for (auto& kv : sub->spectra16m) (void)hipFree(kv.second);
            delete sub;
        }
        for (auto& kv : h->spectra) (void)hipFree(kv.second);
        for (auto& kv : h->spectra16) (void)hipFree(kv.second);
        for (auto& kv : h->spectra16m) (void)hipFree(kv.second);
        if (h->d_needle && h->owns_data) (void)hipFree(h->d_needle);
    }
    delete h;
}

int am_needle_len(const am_needle* h, size_t* n) {
    if (!h || !n) return fail(AM_ERR_INVALID_ARG, "null pointer");
    *n = h->n;
    return AM_OK;
}

int am_needle_inv_autocorr(const am_needle* h, float* out) {
    if (!h || !out) return fail(AM_ERR_INVALID_ARG, "null pointer");
    *out = h->inv_autocorr;
    return AM_OK;
}

int am_correlate_len(size_t w, size_t s, int mode, size_t* out_len) {
    if (!out_len || w == 0 || s == 0 || mode < 0 || mode > 2) return fail(AM_ERR_INVALID_ARG, "bad argument");
    *out_len = mode_len(w, s, mode);
    return AM_OK;
}

static int correlate_impl(const am_needle* hc, const float* within, size_t w, int mode, int scale,
                          float* out, size_t cap, size_t* out_len, bool device_io) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!within || !out_len || w == 0) return fail(AM_ERR_INVALID_ARG, "within must be non-empty");
    if (mode < AM_MODE_FULL || mode > AM_MODE_VALID) return fail(AM_ERR_INVALID_ARG, "bad mode");
    if (scale < AM_SCALE_NONE || scale > AM_SCALE_MY) return fail(AM_ERR_INVALID_ARG, "bad scale");
    const size_t s = h->n;
    const size_t len = mode_len(w, s, mode);
    *out_len = len;
    if (cap < len || !out) return fail(AM_ERR_CAPACITY, "output buffer too small");
    // centered(): start = (full - len) / 2 (audio_matcher.rs:460-464)
    const size_t start = (w + s - 1 - len) / 2;
    const long long lead = (long long)(s - 1) - (long long)start;
    Ctx* c = h->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const float* d_in = within;
    float* d_out = out;
    if (!device_io) {
        if ((rc = c->io_in.ensure(w * sizeof(float)))) return rc;
        if ((rc = c->io_out.ensure(len * sizeof(float)))) return rc;
        AM_HIP(copy_on_stream(c, c->io_in.p, within, w * sizeof(float), hipMemcpyHostToDevice));
        d_in = (const float*)c->io_in.p;
        d_out = (float*)c->io_out.p;
    }
    Opts o = snapshot_opts(h);
    if ((rc = run_correlation(h, o, d_in, (long long)w, lead, d_out, (long long)len, scale_factor(h, scale, w)))) return rc;
    if (o.half) {
        // a half-precision transform can leave f16's range (see match_many): look at the result once and
        // compute it again in f32 if it holds a non-finite value (a bad input is dealt with below)
        const Segment whole{0, (long long)len};
        int flag = 0;
        if ((rc = c->ranges.ensure(sizeof(Segment)))) return rc;
        if ((rc = c->range_flags.ensure(sizeof(int)))) return rc;
        AM_HIP(hipMemcpyAsync(c->ranges.p, &whole, sizeof(Segment), hipMemcpyHostToDevice, c->stream));
        AM_HIP(hipMemsetAsync(c->range_flags.p, 0, sizeof(int), c->stream));
        AM_HIP(launch_nonfinite_ranges(c->stream, d_out, (const Segment*)c->ranges.p, 1, (int*)c->range_flags.p));
        AM_HIP(hipMemcpyAsync(&flag, c->range_flags.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        AM_HIP(hipStreamSynchronize(c->stream));
        if (flag) {
            o.half = 0;
            if ((rc = run_correlation(h, o, d_in, (long long)w, lead, d_out, (long long)len, scale_factor(h, scale, w)))) return rc;
        }
    }
    // A NaN or an infinity in `within` makes every output of the reference's one transform per
    // window NaN (audio_matcher.rs:414-457); overlap-save confines it to the block pairs around
    // it.  Look at the window once and give the reference's answer.
    {
        const Segment whole{0, (long long)w};
        int flag = 0;
        if ((rc = c->ranges.ensure(sizeof(Segment)))) return rc;
        if ((rc = c->range_flags.ensure(sizeof(int)))) return rc;
        AM_HIP(hipMemcpyAsync(c->ranges.p, &whole, sizeof(Segment), hipMemcpyHostToDevice, c->stream));
        AM_HIP(hipMemsetAsync(c->range_flags.p, 0, sizeof(int), c->stream));
        AM_HIP(launch_nonfinite_ranges(c->stream, d_in, (const Segment*)c->ranges.p, 1, (int*)c->range_flags.p));
        AM_HIP(hipMemcpyAsync(&flag, c->range_flags.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        AM_HIP(hipStreamSynchronize(c->stream));
        if (flag) AM_HIP(hipMemsetD32Async((hipDeviceptr_t)d_out, 0x7FC00000, len, c->stream));
    }
    AM_HIP(hipStreamSynchronize(c->stream));
    if (!device_io) AM_HIP(copy_on_stream(c, out, d_out, len * sizeof(float), hipMemcpyDeviceToHost));
    return AM_OK;
}

int am_correlate(const am_needle* h, const float* within, size_t w, int mode, int scale,
                 float* out, size_t cap, size_t* out_len) {
    return correlate_impl(h, within, w, mode, scale, out, cap, out_len, false);
}

int am_correlate_device(const am_needle* h, const float* d_within, size_t w, int mode, int scale,
                        float* d_out, size_t cap, size_t* out_len) {
    return correlate_impl(h, d_within, w, mode, scale, d_out, cap, out_len, true);
}

int am_match_device(const am_needle* hc, const float* d_haystack, size_t len,
                    const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!d_haystack || !p || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (len == 0) { *n_out = 0; return AM_OK; }
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    const void* src = d_haystack;
    return match_many(h, &src, &len, 1, p, out, cap, n_out);
}

int am_match(const am_needle* hc, const float* haystack, size_t len,
             const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!haystack || !p || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (len == 0) { *n_out = 0; return AM_OK; }
    Ctx* c = h->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if ((rc = c->io_in.ensure(len * sizeof(float)))) return rc;
    AM_HIP(copy_on_stream(c, c->io_in.p, haystack, len * sizeof(float), hipMemcpyHostToDevice));
    const void* d_in = c->io_in.p;
    return match_many(h, &d_in, &len, 1, p, out, cap, n_out);
}

int am_match_batch_device(const am_needle* hc, const float* const* d_haystacks, const size_t* lens,
                          size_t n_hay, const am_match_params* p,
                          am_peak* out, size_t cap_per_hay, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!d_haystacks || !lens || !p || !n_out || (!out && cap_per_hay)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    if (n_hay == 0) return AM_OK;
    return match_many(h, reinterpret_cast<const void* const*>(d_haystacks), lens, n_hay, p, out, cap_per_hay, n_out);
}

int am_match_multi_device(const am_needle* const* needles, size_t n_needles, const float* d_haystack, size_t len,
                          const am_match_params* p, am_peak* out, size_t cap_per_needle, size_t* n_out) {
    if (!needles || n_needles == 0 || !d_haystack || !p || !n_out || (!out && cap_per_needle))
        return fail(AM_ERR_INVALID_ARG, "null pointer");
    for (size_t k = 0; k < n_needles; ++k)
        if (!needles[k] || !needles[k]->ctx) return fail(AM_ERR_INVALID_ARG, "null needle handle");
    am_needle* h0 = const_cast<am_needle*>(needles[0]);
    int rc = check_needle(h0);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(h0->ctx->mu);
    const void* src = d_haystack;
    return match_multi_many(const_cast<am_needle* const*>(needles), n_needles, &src, &len, 1, AM_FMT_F32_MONO, p, out, cap_per_needle, n_out);
}

int am_match_multi_batch_device(const am_needle* const* needles, size_t n_needles, const void* const* d_haystacks,
                                const size_t* lens, size_t n_hay, int sample_format, const am_match_params* p,
                                am_peak* out, size_t cap_per_pair, size_t* n_out) {
    if (!needles || n_needles == 0 || !d_haystacks || !lens || !p || !n_out || (!out && cap_per_pair))
        return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (sample_format != AM_FMT_F32_MONO && sample_format != AM_FMT_S16_STEREO) return fail(AM_ERR_INVALID_ARG, "bad sample format");
    for (size_t k = 0; k < n_needles; ++k)
        if (!needles[k] || !needles[k]->ctx) return fail(AM_ERR_INVALID_ARG, "null needle handle");
    am_needle* h0 = const_cast<am_needle*>(needles[0]);
    int rc = check_needle(h0);
    if (rc) return rc;
    if (n_hay == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(h0->ctx->mu);
    return match_multi_many(const_cast<am_needle* const*>(needles), n_needles, d_haystacks, lens, n_hay, sample_format, p,
                            out, cap_per_pair, n_out);
}

// ---- streaming ingest -------------------------------------------------------------------
int am_match_stream_begin(const am_needle* hc, int sample_format, size_t expected_len, const am_match_params* p, am_stream** out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!p || !out) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (sample_format != AM_FMT_F32_MONO && sample_format != AM_FMT_S16_STEREO) return fail(AM_ERR_INVALID_ARG, "bad sample format");
    if (p->chunk == 0) return fail(AM_ERR_INVALID_ARG, "chunk must be > 0");
    if (p->scale < AM_SCALE_NONE || p->scale > AM_SCALE_MY) return fail(AM_ERR_INVALID_ARG, "bad scale");
    Ctx* c = h->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    am_stream* st = new am_stream();
    st->h = h; st->fmt = sample_format; st->p = *p;
    if (hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&st->copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&st->staged[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&st->staged[1], hipEventDisableTiming) != hipSuccess) {
        am_match_stream_destroy(st);
        return fail(AM_ERR_HIP, "hipStreamCreate(stream ingest)");
    }
    // the size hint of the reference's iterator (mp3_duration x sample rate, matcher/mod.rs:77-83) may be off
    // by a little: leave room, so that a slightly longer file does not force a new layout
    st->cap = expected_len ? expected_len + expected_len / 64 + 65536 : 0;
    if (st->cap) {
        if ((rc = st->hay.ensure(st->cap * 4)) || (rc = stream_layout(st))) { const std::string keep = t_err; am_match_stream_destroy(st); t_err = keep; return rc; }
    }
    *out = st;
    return AM_OK;
}

// the current staging slot goes to the device (asynchronously); the other slot becomes current
static int stream_flush_slot(am_stream* st) {
    if (st->fill == 0) return AM_OK;
    const int b = st->cur;
    hipError_t e = hipMemcpyAsync(static_cast<char*>(st->hay.p) + st->sent * 4, st->stage[b].p, st->fill * 4, hipMemcpyHostToDevice, st->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(st->staged[b], st->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(st->copied, st->copy_stream);
    if (e != hipSuccess) { st->failed = true; return hip_fail(e, "stream ingest: copy"); }
    st->stage_busy[b] = true;
    st->sent += st->fill;
    st->fill = 0;
    st->cur = b ^ 1;
    return AM_OK;
}

// K1 / K2 / K3 for every block pair whose samples are on their way to the device (st->sent)
static int stream_launch_ready_pairs(am_stream* st) {
    if (!st->early) return AM_OK;
    am_needle* h = st->h;
    Ctx* c = h->ctx;
    // pairs whose two blocks lie completely inside what has arrived: K1 reads [2q hop, (2q + 1) hop + N)
    const Geometry& g = st->geo;
    const long long have = (long long)st->sent;
    long long ready = have >= g.hop + g.N ? ((have - g.N) / g.hop - 1) / 2 + 1 : 0;
    ready = std::min(ready, g.npairs);
    if (ready - st->pairs_done < 1) return AM_OK;
    const Opts o = snapshot_opts(h);
    Geometry now{};
    int rc = plan_geometry(h->n, (long long)(st->cap - h->n + 1), o, &now);
    if (rc) return rc;
    if (now.logN != g.logN || now.hop != g.hop) {   // an option changed under the stream: start over at finish
        st->early = false; st->pairs_done = 0;
        return AM_OK;
    }
    ScanRequest scan{};
    scan.margin = (!o.dense && st->p.min_prominence > 0.f) ? 0.5f * st->p.min_prominence : -1.0f;
    scan.hist_min = h->hist_min(st->p.scale == AM_SCALE_LIB ? 1 : 0);
    scan.seg_c = (long long)st->p.chunk;
    scan.seg_d = (long long)(st->p.chunk + st->p.overlap) - (long long)h->n;
    if (st->pairs_done > 0 && scan.margin != st->margin) {
        // "dense_scores" changed between two pushes: the early pairs were written under another rule than the
        // rest would be -- start over at finish
        st->early = false; st->pairs_done = 0;
        return AM_OK;
    }
    st->margin = scan.margin;
    scan.ext_stats32 = &st->stats32; scan.ext_side = &st->side;
    scan.side_nblocks = g.nblocks;
    scan.range_a = st->pairs_done * 2 * g.hop;
    scan.range_b = ready * 2 * g.hop;
    AM_HIP(hipStreamWaitEvent(c->stream, st->copied, 0));   // the kernels read what has been copied so far
    rc = run_correlation(h, o, st->hay.p, (long long)st->cap, 0, (float*)st->scores.p, (long long)(st->cap - h->n + 1),
                         scale_factor(h, st->p.scale, 1), &scan, st->fmt);
    if (rc) { st->failed = true; return rc; }
    st->pairs_done = ready;
    return AM_OK;
}

int am_match_stream_push(am_stream* st, const void* samples, size_t n) {
    if (!st || !st->h || (!samples && n)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (st->failed) return fail(AM_ERR_INVALID_ARG, "stream is in a failed state: destroy it");
    if (n == 0) return AM_OK;
    am_needle* h = st->h;
    int rc = check_needle(h);
    if (rc) return rc;
    Ctx* c = h->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (st->len + n > st->cap) {
        // longer than announced: a larger buffer (contents moved on the device) and a new provisional layout;
        // the pairs computed so far are computed again (the layout may differ)
        const size_t want = std::max(st->len + n, st->cap * 2 + 65536);
        DevBuf bigger;
        if ((rc = bigger.ensure(want * 4))) { st->failed = true; return rc; }
        hipError_t e = hipStreamSynchronize(st->copy_stream);
        if (e == hipSuccess && st->sent) e = copy_on_stream(c, bigger.p, st->hay.p, st->sent * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { bigger.release(); st->failed = true; return hip_fail(e, "stream ingest: grow"); }
        st->hay.release();
        st->hay = bigger;
        st->cap = want;
        if ((rc = stream_layout(st))) { st->failed = true; return rc; }
    }
    if (n >= am_stream::kDirectElems) {
        // a large piece: what the ring holds goes first (order), then one copy straight from the caller's buffer;
        // the caller may reuse `samples` as soon as this returns, so that copy is waited for
        if ((rc = stream_flush_slot(st))) return rc;
        hipError_t e = hipMemcpyAsync(static_cast<char*>(st->hay.p) + st->sent * 4, samples, n * 4, hipMemcpyHostToDevice, st->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(st->copied, st->copy_stream);
        if (e != hipSuccess) { st->failed = true; return hip_fail(e, "stream ingest: copy"); }
        st->sent += n;
        st->len += n;
        rc = stream_launch_ready_pairs(st);
        AM_HIP(hipStreamSynchronize(st->copy_stream));
        return rc;
    }
    // a small piece: a host memcpy into the staging ring; full slots leave asynchronously
    const char* src = static_cast<const char*>(samples);
    size_t left = n;
    bool flushed = false;
    while (left) {
        const int b = st->cur;
        if (st->fill == 0) {
            if (!st->stage[b].p && (rc = st->stage[b].ensure(am_stream::kStageElems * 4))) { st->failed = true; return rc; }
            if (st->stage_busy[b]) {   // (the copy that last left this slot: two slots ago)
                AM_HIP(hipEventSynchronize(st->staged[b]));
                st->stage_busy[b] = false;
            }
        }
        const size_t take = std::min(left, am_stream::kStageElems - st->fill);
        memcpy(static_cast<char*>(st->stage[b].p) + st->fill * 4, src, take * 4);
        st->fill += take; st->len += take;
        src += take * 4; left -= take;
        if (st->fill == am_stream::kStageElems) {
            if ((rc = stream_flush_slot(st))) return rc;
            flushed = true;
        }
    }
    return flushed ? stream_launch_ready_pairs(st) : AM_OK;
}

int am_match_stream_finish(am_stream* st, am_peak* out, size_t cap, size_t* n_out) {
    if (!st || !st->h || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (st->failed) return fail(AM_ERR_INVALID_ARG, "stream is in a failed state: destroy it");
    am_needle* h = st->h;
    int rc = check_needle(h);
    if (rc) return rc;
    Ctx* c = h->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    *n_out = 0;
    const size_t len = st->len;
    rc = AM_OK;
    if (len) {
        if ((rc = stream_flush_slot(st))) return rc;          // what the staging ring still holds
        AM_HIP(hipStreamWaitEvent(c->stream, st->copied, 0));
        const void* src = st->hay.p;
        StreamPre pre{(float*)st->scores.p, &st->stats32, &st->side, st->pairs_done, st->geo.nblocks};
        bool use_pre = st->early && st->pairs_done > 0;
        if (use_pre) {
            // the layout the whole haystack gets must be the one the early pairs were computed in (the side
            // buffer keeps the offsets of the announced length: StreamPre::layout_nblocks), and so must the rule
            // by which raw scores are written: with another margin (dense_scores switched, or a prominence bound
            // that is no longer positive) the pick would read runs the early pairs never wrote
            Geometry fin{};
            const Opts o = snapshot_opts(h);
            const float margin = (st->p.scale != AM_SCALE_MY && !o.dense && st->p.min_prominence > 0.f) ? 0.5f * st->p.min_prominence : -1.0f;
            if (len < h->n || plan_geometry(h->n, (long long)(len - h->n + 1), o, &fin) || fin.logN != st->geo.logN || fin.hop != st->geo.hop ||
                fin.nblocks > st->geo.nblocks || margin != st->margin)
                use_pre = false;
        }
        rc = match_many(h, &src, &len, 1, &st->p, out, cap, n_out, st->fmt, 0, 1, true, use_pre ? &pre : nullptr);
    }
    // ready for the next file of the same (announced) size; a stream that had to give up its early pairs (an
    // option changed under it) starts afresh
    st->len = 0; st->sent = 0; st->pairs_done = 0;
    if (!st->early && st->cap) (void)stream_layout(st);
    return rc;
}

void am_match_stream_destroy(am_stream* st) {
    if (!st) return;
    if (st->h && st->h->ctx) {
        std::lock_guard<std::recursive_mutex> lk(st->h->ctx->mu);
        (void)hipSetDevice(st->h->ctx->device);
        if (st->copy_stream) (void)hipStreamSynchronize(st->copy_stream);
        (void)hipStreamSynchronize(st->h->ctx->stream);
        st->hay.release(); st->scores.release(); st->stats32.release(); st->side.release();
    }
    if (st->copied) (void)hipEventDestroy(st->copied);
    for (int b = 0; b < 2; ++b) {
        if (st->staged[b]) (void)hipEventDestroy(st->staged[b]);
        if (st->stage[b].p) (void)hipHostFree(st->stage[b].p);
    }
    if (st->copy_stream) (void)hipStreamDestroy(st->copy_stream);
    delete st;
}

// ---- the same three entry points on interleaved i16 stereo PCM: the down-mix of
// mp3_reader.rs:28-37 happens inside K1's loads, so the haystack is read once ----
int am_match_pcm16_device(const am_needle* hc, const int16_t* d_interleaved, size_t frames,
                          const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!d_interleaved || !p || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (frames == 0) { *n_out = 0; return AM_OK; }
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    const void* src = d_interleaved;
    return match_many(h, &src, &frames, 1, p, out, cap, n_out, 1);
}

int am_match_pcm16(const am_needle* hc, const int16_t* interleaved, size_t frames,
                   const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!interleaved || !p || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (frames == 0) { *n_out = 0; return AM_OK; }
    Ctx* c = h->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if ((rc = c->io_in.ensure(frames * 2 * sizeof(int16_t)))) return rc;
    AM_HIP(copy_on_stream(c, c->io_in.p, interleaved, frames * 2 * sizeof(int16_t), hipMemcpyHostToDevice));
    const void* d_in = c->io_in.p;
    return match_many(h, &d_in, &frames, 1, p, out, cap, n_out, 1);
}

int am_match_pcm16_batch_device(const am_needle* hc, const int16_t* const* d_interleaved, const size_t* frames,
                                size_t n_hay, const am_match_params* p,
                                am_peak* out, size_t cap_per_hay, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!d_interleaved || !frames || !p || !n_out || (!out && cap_per_hay)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    if (n_hay == 0) return AM_OK;
    return match_many(h, reinterpret_cast<const void* const*>(d_interleaved), frames, n_hay, p, out, cap_per_hay, n_out, 1);
}

int am_needle_create_pcm16(int device, const int16_t* interleaved, size_t frames, am_needle** out) {
    if (!interleaved || !out || frames == 0) return fail(AM_ERR_INVALID_ARG, "needle must be non-empty");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if ((rc = c->io_in.ensure(frames * 2 * sizeof(int16_t)))) return rc;
    AM_HIP(copy_on_stream(c, c->io_in.p, interleaved, frames * 2 * sizeof(int16_t), hipMemcpyHostToDevice));
    float* d = nullptr;
    AM_HIP(hipMalloc((void**)&d, frames * sizeof(float)));
    hipError_t e = launch_pcm_downmix(c->stream, (const int16_t*)c->io_in.p, (long long)frames, d);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "needle down-mix"); }
    return create_needle_common(c, d, frames, out);
}

int am_find_peaks(int device, const float* scores, size_t n, float min_prominence,
                  uint64_t min_distance, am_peak* out, size_t cap, size_t* n_out) {
    if (!scores || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    *n_out = 0;
    if (n == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if ((rc = c->io_in.ensure(n * sizeof(float)))) return rc;
    AM_HIP(copy_on_stream(c, c->io_in.p, scores, n * sizeof(float), hipMemcpyHostToDevice));
    std::vector<am_peak> all;
    if ((rc = find_peaks_host_array(c, (const float*)c->io_in.p, (long long)n, min_prominence,
                                    (long long)min_distance, all))) return rc;
    *n_out = all.size();
    for (size_t i = 0; i < all.size() && i < cap; ++i) out[i] = all[i];
    if (all.size() > cap) return fail(AM_ERR_CAPACITY, "peak output buffer too small");
    return AM_OK;
}

int am_pcm_s16_stereo_to_mono_device(int device, const int16_t* d_in, size_t frames, float* d_out) {
    if (!d_in || !d_out) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (frames == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    AM_HIP(launch_pcm_downmix(c->stream, d_in, (long long)frames, d_out));
    AM_HIP(hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_pcm_s16_stereo_to_mono(int device, const int16_t* interleaved, size_t frames, float* out) {
    if (!interleaved || !out) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (frames == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if ((rc = c->io_in.ensure(frames * 2 * sizeof(int16_t)))) return rc;
    if ((rc = c->io_out.ensure(frames * sizeof(float)))) return rc;
    AM_HIP(copy_on_stream(c, c->io_in.p, interleaved, frames * 2 * sizeof(int16_t), hipMemcpyHostToDevice));
    AM_HIP(launch_pcm_downmix(c->stream, (const int16_t*)c->io_in.p, (long long)frames, (float*)c->io_out.p));
    AM_HIP(hipStreamSynchronize(c->stream));
    AM_HIP(copy_on_stream(c, out, c->io_out.p, frames * sizeof(float), hipMemcpyDeviceToHost));
    return AM_OK;
}

int am_device_malloc(int device, size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(AM_ERR_INVALID_ARG, "bad argument");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    AM_HIP(hipMalloc(out, bytes));
    return AM_OK;
}
int am_device_free(int device, void* p) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (p) AM_HIP(hipFree(p));
    return AM_OK;
}
// Pinned host memory for the buffers a host hands to am_match / am_match_stream_push / am_pool_match_*: the
// copy engines read it directly (no bounce buffer in the runtime, no page faults), which is what lets N copier
// threads feed N devices side by side.  Portable: usable from every device's context.
int am_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(AM_ERR_INVALID_ARG, "bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(AM_ERR_NO_DEVICE, "no HIP device available");
    AM_HIP(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return AM_OK;
}
int am_host_free(void* p) {
    if (p) AM_HIP(hipHostFree(p));
    return AM_OK;
}
int am_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return fail(AM_ERR_INVALID_ARG, "bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(AM_ERR_NO_DEVICE, "no HIP device available");
    AM_HIP(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return AM_OK;
}
int am_host_unregister(void* p) {
    if (!p) return fail(AM_ERR_INVALID_ARG, "null pointer");
    AM_HIP(hipHostUnregister(p));
    return AM_OK;
}

int am_memcpy_h2d(int device, void* d_dst, const void* src, size_t bytes) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    AM_HIP(copy_on_stream(c, d_dst, src, bytes, hipMemcpyHostToDevice));
    return AM_OK;
}
int am_memcpy_d2h(int device, void* dst, const void* d_src, size_t bytes) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    AM_HIP(copy_on_stream(c, dst, d_src, bytes, hipMemcpyDeviceToHost));
    return AM_OK;
}
int am_device_synchronize(int device) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    AM_HIP(hipDeviceSynchronize());
    return AM_OK;
}

int am_synth_uniform_device(int device, float* d_out, uint32_t seed, uint32_t stream,
                            uint64_t first, size_t n, float amp) {
    if (!d_out) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (n == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    AM_HIP(launch_synth(c->stream, d_out, seed, stream, first, (long long)n, amp));
    AM_HIP(hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_axpy_device(int device, float* d_dst, const float* d_src, size_t n, float gain) {
    if (!d_dst || !d_src) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (n == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    AM_HIP(launch_axpy(c->stream, d_dst, d_src, (long long)n, gain));
    AM_HIP(hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_synth_pcm16_stereo_device(int device, int16_t* d_out, uint32_t seed, uint32_t stream, uint64_t first, size_t frames, float amp) {
    if (!d_out) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (frames == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    AM_HIP(launch_synth_pcm16(c->stream, d_out, seed, stream, first, (long long)frames, amp));
    AM_HIP(hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_add_pcm16_device(int device, int16_t* d_dst, const int16_t* d_src, size_t frames) {
    if (!d_dst || !d_src) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    if (frames == 0) return AM_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    AM_HIP(launch_add_pcm16(c->stream, d_dst, d_src, (long long)frames));
    AM_HIP(hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto& kv : g_ctx) {
        Ctx* c = kv.second;
        std::lock_guard<std::recursive_mutex> lk2(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        if (c->stream_tail) (void)hipStreamSynchronize(c->stream_tail);
        for (DevBuf* b : {&c->work, &c->work2, &c->scores, &c->stats, &c->stats32, &c->wflags, &c->segs,
                          &c->scores_b, &c->stats_b, &c->stats32_b, &c->wflags_b, &c->peaks_b, &c->work_b, &c->redo_pairs[0], &c->redo_pairs[1],
                          &c->peaks, &c->io_in, &c->io_out, &c->sum, &c->arena_cur, &c->wide_ctl, &c->wide_list, &c->wide_tiles})
            b->release();
        if (c->pinned.p) { (void)hipHostFree(c->pinned.p); c->pinned.p = nullptr; c->pinned.cap = 0; }
        if (c->hdr.p) { (void)hipHostFree(c->hdr.p); c->hdr.p = nullptr; c->hdr.cap = 0; }
        if (c->spill.p) { (void)hipHostFree(c->spill.p); c->spill.p = nullptr; c->spill.cap = 0; }
        if (c->badflag.p) { (void)hipHostFree(c->badflag.p); c->badflag.p = nullptr; c->badflag.cap = 0; }
        if (c->failcnt.p) { (void)hipHostFree(c->failcnt.p); c->failcnt.p = nullptr; c->failcnt.cap = 0; }
        c->ranges.release(); c->range_flags.release(); c->big.release();
        c->work_tail.release(); c->tail_scores.release(); c->tail_stats.release(); c->work_tail2.release();
        for (int i = 0; i < 2 * kMaxNeedleGroup; ++i) { c->grp_scores[i].release(); c->grp_stats32[i].release(); c->grp_wflags[i].release(); }
        for (int i = 0; i < kMaxNeedleGroup; ++i) c->grp_stats[i].release();
        c->segs_resident.clear();
        for (auto& pk : c->plans) { if (pk.second.tables) (void)hipFree(pk.second.tables); if (pk.second.mf) (void)hipFree(pk.second.mf); }
        c->plans.clear();
        for (auto& r : c->pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        c->pending.clear();
        for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
        c->pool.clear();
    }
    return AM_OK;
}


// ---- pool ---------------------------------------------------------------------
int am_shard_plan(size_t n_items, size_t n_shards, size_t shard, size_t* first, size_t* stride, size_t* count) {
    if (!first || !stride || !count || n_shards == 0 || shard >= n_shards) return fail(AM_ERR_INVALID_ARG, "bad shard");
    *first = shard;
    *stride = n_shards;
    *count = n_items > shard ? (n_items - shard + n_shards - 1) / n_shards : 0;
    return AM_OK;
}

static int pool_create_common(const float* const* needles, size_t n_needles, size_t n, const int* devices, size_t n_dev, am_pool** out) {
    if (!needles || !out || n == 0 || n_needles == 0) return fail(AM_ERR_INVALID_ARG, "needle must be non-empty");
    for (size_t j = 0; j < n_needles; ++j) if (!needles[j]) return fail(AM_ERR_INVALID_ARG, "null needle");
    std::vector<int> devs;
    if (devices) {
        if (n_dev == 0) return fail(AM_ERR_INVALID_ARG, "empty device list");
        devs.assign(devices, devices + n_dev);
    } else {
        int k = 0;
        if (hipGetDeviceCount(&k) != hipSuccess || k <= 0) return fail(AM_ERR_NO_DEVICE, "no HIP device available");
        for (int d = 0; d < k; ++d) devs.push_back(d);
    }
    am_pool* pool = new am_pool();
    pool->slots.resize(devs.size());
    for (size_t i = 0; i < devs.size(); ++i) {
        am_pool::Slot& sl = pool->slots[i];
        sl.device = devs[i];
        int rc = AM_OK;
        for (size_t j = 0; j < n_needles && rc == AM_OK; ++j) {
            am_needle* h = nullptr;
            rc = am_needle_create(devs[i], needles[j], n, &h);
            if (rc == AM_OK) sl.needles.push_back(h);
        }
        if (rc == AM_OK) sl.needle = sl.needles[0];
        if (rc == AM_OK && hipStreamCreateWithFlags(&sl.copy_stream, hipStreamNonBlocking) != hipSuccess)
            rc = fail(AM_ERR_HIP, "hipStreamCreate(pool copy stream)");
        if (rc) { const std::string keep = t_err; am_pool_destroy(pool); t_err = keep; return rc; }
    }
    *out = pool;
    return AM_OK;
}

int am_pool_create(const float* needle, size_t n, const int* devices, size_t n_dev, am_pool** out) {
    if (!needle) return fail(AM_ERR_INVALID_ARG, "needle must be non-empty");
    return pool_create_common(&needle, 1, n, devices, n_dev, out);
}

int am_pool_create_multi(const float* const* needles, size_t n_needles, size_t n, const int* devices, size_t n_dev, am_pool** out) {
    return pool_create_common(needles, n_needles, n, devices, n_dev, out);
}

void am_pool_destroy(am_pool* pool) {
    if (!pool) return;
    for (am_pool::Slot& sl : pool->slots) {
        if (sl.device >= 0) (void)hipSetDevice(sl.device);
        if (sl.copy_stream) { (void)hipStreamSynchronize(sl.copy_stream); (void)hipStreamDestroy(sl.copy_stream); }
        for (void* r : sl.ring) if (r) (void)hipFree(r);
        for (am_needle* h : sl.needles) am_needle_destroy(h);
    }
    delete pool;
}

int am_pool_size(const am_pool* pool, size_t* n_dev) {
    if (!pool || !n_dev) return fail(AM_ERR_INVALID_ARG, "null pointer");
    *n_dev = pool->slots.size();
    return AM_OK;
}

int am_pool_slot(const am_pool* pool, size_t slot, int* device, const am_needle** needle) {
    if (!pool || slot >= pool->slots.size()) return fail(AM_ERR_INVALID_ARG, "bad pool slot");
    if (device) *device = pool->slots[slot].device;
    if (needle) *needle = pool->slots[slot].needle;
    return AM_OK;
}

namespace {

// What a pool call runs per haystack: one needle (match_many; out holds cap slots per haystack) or
// every needle of the pool (match_multi_many; cap slots per (haystack, needle) pair, slot k * nn + j).
struct PoolJob {
    bool multi;
    int fmt;   // AM_FMT_*: one f32 mono sample and one i16 stereo frame are both 4 bytes
};

// A resident haystack must live on the device of the slot that matches it (haystack k on slot k mod n_dev):
// the kernels of that device would otherwise read it over xGMI, or fault.  Ask the runtime instead of
// trusting the caller.
int check_resident(const void* ptr, int device, size_t index) {
    hipPointerAttribute_t attr{};
    const hipError_t e = hipPointerGetAttributes(&attr, ptr);
    char buf[200];
    if (e != hipSuccess) {
        (void)hipGetLastError();
        snprintf(buf, sizeof(buf), "haystack %zu: not a device pointer the runtime knows (%s)", index, hipGetErrorString(e));
        return fail(AM_ERR_INVALID_ARG, buf);
    }
    if (attr.type == hipMemoryTypeManaged) return AM_OK;
    if (attr.type != hipMemoryTypeDevice) {
        snprintf(buf, sizeof(buf), "haystack %zu: host memory passed to a _device entry point", index);
        return fail(AM_ERR_INVALID_ARG, buf);
    }
    if (attr.device != device) {
        snprintf(buf, sizeof(buf), "haystack %zu lives on device %d but its pool slot runs on device %d (haystack k belongs on slot k mod n_dev)",
                 index, attr.device, device);
        return fail(AM_ERR_INVALID_ARG, buf);
    }
    return AM_OK;
}

int slot_match(am_pool::Slot& sl, const PoolJob& job, const void* const* ptrs, const size_t* ln, size_t count,
               const am_match_params* p, am_peak* out, size_t cap, size_t* n_out, size_t first, size_t stride) {
    am_needle* h = sl.needle;
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    if (job.multi)
        return match_multi_many(sl.needles.data(), sl.needles.size(), ptrs, ln, count, job.fmt, p, out, cap, n_out, first, stride);
    return match_many(h, ptrs, ln, count, p, out, cap, n_out, job.fmt, first, stride);
}

// resident haystacks: the slot's shard goes through the matcher as one batch
int slot_run_device(am_pool::Slot& sl, const PoolJob& job, size_t slot, size_t nslots, const void* const* d_hays, const size_t* lens,
                    size_t n_hay, const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    size_t first, stride, count;
    am_shard_plan(n_hay, nslots, slot, &first, &stride, &count);
    if (count == 0) return AM_OK;
    std::vector<const void*> ptrs(count);
    std::vector<size_t> ln(count);
    for (size_t i = 0; i < count; ++i) { ptrs[i] = d_hays[first + i * stride]; ln[i] = lens[first + i * stride]; }
    int rc = check_needle(sl.needle);
    if (rc) return rc;
    for (size_t i = 0; i < count; ++i)
        if (ptrs[i] && ln[i] && (rc = check_resident(ptrs[i], sl.device, first + i * stride))) return rc;
    return slot_match(sl, job, ptrs.data(), ln.data(), count, p, out, cap, n_out, first, stride);
}

// host haystacks: a copier thread fills the two-slot ring one haystack ahead of the matcher
int slot_run_host(am_pool::Slot& sl, const PoolJob& job, size_t slot, size_t nslots, const void* const* hays, const size_t* lens,
                  size_t n_hay, const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    size_t first, stride, count;
    am_shard_plan(n_hay, nslots, slot, &first, &stride, &count);
    if (count == 0) return AM_OK;
    am_needle* h = sl.needle;
    int rc = check_needle(h);   // hipSetDevice for this thread
    if (rc) return rc;
    size_t max_len = 0;
    for (size_t i = 0; i < count; ++i) if (hays[first + i * stride]) max_len = std::max(max_len, lens[first + i * stride]);
    if (max_len * 4 > sl.ring_cap) {
        for (void*& r : sl.ring) { if (r) (void)hipFree(r); r = nullptr; }
        sl.ring_cap = 0;
        for (void*& r : sl.ring) {
            hipError_t e = hipMalloc(&r, max_len * 4);
            if (e != hipSuccess) { r = nullptr; return hip_fail(e, "hipMalloc(pool ring)"); }
        }
        sl.ring_cap = max_len * 4;
    }
    std::mutex m;
    std::condition_variable cv;
    bool ready[2] = {false, false};
    hipError_t copy_err = hipSuccess;
    bool stop = false;
    std::thread copier([&] {
        (void)hipSetDevice(sl.device);
        for (size_t i = 0; i < count; ++i) {
            const int b = (int)(i & 1);
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !ready[b] || stop; });
                if (stop) return;
            }
            const size_t k = first + i * stride;
            hipError_t e = hipSuccess;
            if (hays[k] && lens[k]) {
                e = hipMemcpyAsync(sl.ring[b], hays[k], lens[k] * 4, hipMemcpyHostToDevice, sl.copy_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(sl.copy_stream);
            }
            std::lock_guard<std::mutex> lk(m);
            if (e != hipSuccess) { copy_err = e; stop = true; cv.notify_all(); return; }
            ready[b] = true;
            cv.notify_all();
        }
    });
    int worst = AM_OK;
    for (size_t i = 0; i < count; ++i) {
        const int b = (int)(i & 1);
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return ready[b] || stop; });
            if (stop) break;
        }
        const size_t k = first + i * stride;
        const void* src = (hays[k] && lens[k]) ? sl.ring[b] : nullptr;
        rc = slot_match(sl, job, &src, &lens[k], 1, p, out, cap, n_out, k, 1);
        {
            std::lock_guard<std::mutex> lk(m);
            ready[b] = false;
            if (rc != AM_OK && rc != AM_ERR_CAPACITY) stop = true;
            cv.notify_all();
        }
        if (rc == AM_ERR_CAPACITY) worst = rc;
        else if (rc) { worst = rc; break; }
    }
    copier.join();
    if (copy_err != hipSuccess) return hip_fail(copy_err, "host-to-device copy (pool)");
    return worst;
}

int pool_run(am_pool* pool, const PoolJob& job, const void* const* hays, const size_t* lens, size_t n_hay, const am_match_params* p,
             am_peak* out, size_t cap, size_t* n_out, bool host) {
    if (!pool || !hays || !lens || !p || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (job.fmt != AM_FMT_F32_MONO && job.fmt != AM_FMT_S16_STEREO) return fail(AM_ERR_INVALID_ARG, "bad sample format");
    std::lock_guard<std::mutex> lk(pool->mu);
    const size_t nslots = pool->slots.size();
    const size_t nn = pool->slots.empty() ? 0 : pool->slots[0].needles.size();
    if (!job.multi && nn != 1)
        return fail(AM_ERR_INVALID_ARG, "this pool holds several needles: use am_pool_match_multi_batch*");
    for (size_t k = 0; k < n_hay * (job.multi ? nn : 1); ++k) n_out[k] = 0;
    if (n_hay == 0) return AM_OK;
    std::vector<int> rcs(nslots, AM_OK);
    std::vector<std::string> errs(nslots);
    std::vector<std::thread> threads;
    for (size_t s = 0; s < nslots; ++s)
        threads.emplace_back([&, s] {
            rcs[s] = host ? slot_run_host(pool->slots[s], job, s, nslots, hays, lens, n_hay, p, out, cap, n_out)
                          : slot_run_device(pool->slots[s], job, s, nslots, hays, lens, n_hay, p, out, cap, n_out);
            if (rcs[s]) errs[s] = t_err;   // the error string is thread-local: hand it to the caller's thread
        });
    for (std::thread& th : threads) th.join();
    int worst = AM_OK;
    for (size_t s = 0; s < nslots; ++s) {
        if (rcs[s] == AM_OK) continue;
        if (worst == AM_OK || worst == AM_ERR_CAPACITY) { worst = rcs[s]; t_err = errs[s]; }
    }
    return worst;
}

}  // namespace

int am_pool_match_batch(am_pool* pool, const float* const* haystacks, const size_t* lens, size_t n_hay,
                        const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out) {
    return pool_run(pool, PoolJob{false, AM_FMT_F32_MONO}, reinterpret_cast<const void* const*>(haystacks), lens, n_hay, p, out, cap_per_hay, n_out, true);
}

int am_pool_match_batch_device(am_pool* pool, const float* const* d_haystacks, const size_t* lens, size_t n_hay,
                               const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out) {
    return pool_run(pool, PoolJob{false, AM_FMT_F32_MONO}, reinterpret_cast<const void* const*>(d_haystacks), lens, n_hay, p, out, cap_per_hay, n_out, false);
}

int am_pool_match_batch_pcm16(am_pool* pool, const int16_t* const* interleaved, const size_t* frames, size_t n_hay,
                              const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out) {
    return pool_run(pool, PoolJob{false, AM_FMT_S16_STEREO}, reinterpret_cast<const void* const*>(interleaved), frames, n_hay, p, out, cap_per_hay, n_out, true);
}

int am_pool_match_batch_pcm16_device(am_pool* pool, const int16_t* const* d_interleaved, const size_t* frames, size_t n_hay,
                                     const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out) {
    return pool_run(pool, PoolJob{false, AM_FMT_S16_STEREO}, reinterpret_cast<const void* const*>(d_interleaved), frames, n_hay, p, out, cap_per_hay, n_out, false);
}

int am_pool_match_multi_batch(am_pool* pool, const void* const* haystacks, const size_t* lens, size_t n_hay, int sample_format,
                              const am_match_params* p, am_peak* out, size_t cap_per_pair, size_t* n_out) {
    return pool_run(pool, PoolJob{true, sample_format}, haystacks, lens, n_hay, p, out, cap_per_pair, n_out, true);
}

int am_pool_match_multi_batch_device(am_pool* pool, const void* const* d_haystacks, const size_t* lens, size_t n_hay, int sample_format,
                                     const am_match_params* p, am_peak* out, size_t cap_per_pair, size_t* n_out) {
    return pool_run(pool, PoolJob{true, sample_format}, d_haystacks, lens, n_hay, p, out, cap_per_pair, n_out, false);
}

// ---- one long haystack over several devices ---------------------------------------------
// calc_chunks fans the windows of ONE haystack out over its workers (audio_matcher.rs:104-131) and sorts and
// filters the union afterwards (:132-140).  The same split here: contiguous window ranges per part, each part's
// buffer reaching to the end of its last window (the overlap tail = the S - 1 halo of SURVEY.md 8e and more),
// the windows of a part matched as one haystack of their own, ONE merge over all parts.
int am_long_plan(size_t len, size_t needle_len, const am_match_params* p, size_t n_parts, size_t part,
                 size_t* first_window, size_t* n_windows, size_t* first_sample, size_t* n_samples) {
    if (!p || !first_window || !n_windows || !first_sample || !n_samples || n_parts == 0 || part >= n_parts || needle_len == 0)
        return fail(AM_ERR_INVALID_ARG, "bad part");
    if (p->chunk == 0) return fail(AM_ERR_INVALID_ARG, "chunk must be > 0");
    // windows that yield scores: i * chunk < len and min(chunk + overlap, len - i * chunk) >= needle_len (make_segments)
    const unsigned long long window = p->chunk + p->overlap;
    size_t nv = 0;
    if (g_opt_tail_window.load(std::memory_order_relaxed)) {   // option "tail_window" = 1: full-length windows only
        if (len >= window && window >= needle_len) nv = (size_t)((len - window) / p->chunk) + 1;
    } else if (len >= needle_len && window >= needle_len) {
        // the last offset whose window is long enough: off <= len - needle_len
        nv = (size_t)((len - needle_len) / p->chunk) + 1;
    }
    const size_t w0 = nv * part / n_parts, w1 = nv * (part + 1) / n_parts;
    *first_window = w0;
    *n_windows = w1 - w0;
    *first_sample = w0 * (size_t)p->chunk;
    *n_samples = 0;
    if (w1 > w0) {
        const unsigned long long end = std::min<unsigned long long>(len, (unsigned long long)(w1 - 1) * p->chunk + window);
        *n_samples = (size_t)(end - (unsigned long long)*first_sample);
    }
    return AM_OK;
}

int am_match_part_device(const am_needle* hc, const void* d_part, size_t n_samples, int sample_format, const am_match_params* p,
                         size_t n_windows, uint64_t first_sample, am_peak* out, size_t cap, size_t* n_out) {
    am_needle* h = const_cast<am_needle*>(hc);
    int rc = check_needle(h);
    if (rc) return rc;
    if (!p || !n_out || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (sample_format != AM_FMT_F32_MONO && sample_format != AM_FMT_S16_STEREO) return fail(AM_ERR_INVALID_ARG, "bad sample format");
    *n_out = 0;
    if (n_windows == 0 || n_samples == 0) return AM_OK;
    if (!d_part) return fail(AM_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    std::vector<am_peak> raw;
    PartSpec part{n_windows, first_sample, 0, n_windows, &raw};
    size_t n = 0;
    if ((rc = match_many(h, &d_part, &n_samples, 1, p, nullptr, 0, &n, sample_format, 0, 1, true, nullptr, &part))) return rc;
    *n_out = raw.size();
    for (size_t i = 0; i < raw.size() && i < cap; ++i) out[i] = raw[i];
    if (raw.size() > cap) return fail(AM_ERR_CAPACITY, "peak output buffer too small");
    return AM_OK;
}

int am_merge_peaks(const am_match_params* p, const am_peak* peaks, size_t n, am_peak* out, size_t cap, size_t* n_out) {
    if (!p || !n_out || (!peaks && n) || (!out && cap)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    std::vector<am_peak> all(peaks, peaks + n);
    return merge_peaks(all, p, snapshot_opts(nullptr).surrounding_from != 0, out, cap, n_out);
}

namespace {

int pool_long(am_pool* pool, const void* host_hay, const void* const* d_parts, size_t len, int fmt, const am_match_params* p,
              am_peak* out, size_t cap, size_t* n_out) {
    if (!pool || !p || !n_out || (!out && cap) || (!host_hay && !d_parts)) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (fmt != AM_FMT_F32_MONO && fmt != AM_FMT_S16_STEREO) return fail(AM_ERR_INVALID_ARG, "bad sample format");
    std::lock_guard<std::mutex> lk(pool->mu);
    *n_out = 0;
    const size_t nslots = pool->slots.size();
    if (nslots == 0) return fail(AM_ERR_INVALID_ARG, "empty pool");
    if (pool->slots[0].needles.size() != 1)
        return fail(AM_ERR_INVALID_ARG, "this pool holds several needles: am_pool_match_long takes a single-needle pool");
    if (len == 0) return AM_OK;
    const size_t s = pool->slots[0].needle->n;
    struct Part { size_t w0, nw, a, n; };
    std::vector<Part> parts(nslots);
    size_t total_windows = 0;
    for (size_t i = 0; i < nslots; ++i) {
        int rc = am_long_plan(len, s, p, nslots, i, &parts[i].w0, &parts[i].nw, &parts[i].a, &parts[i].n);
        if (rc) return rc;
        total_windows += parts[i].nw;
    }
    const Hooks hooks = snapshot_hooks();
    if (hooks.fn) hooks.fn(hooks.user, 0, 0, total_windows);
    std::vector<std::vector<am_peak>> raw(nslots);
    std::vector<int> rcs(nslots, AM_OK);
    std::vector<std::string> errs(nslots);
    std::vector<std::thread> threads;
    for (size_t i = 0; i < nslots; ++i)
        threads.emplace_back([&, i] {
            const Part& pt = parts[i];
            if (pt.nw == 0) return;
            am_pool::Slot& sl = pool->slots[i];
            int rc = check_needle(sl.needle);   // hipSetDevice for this thread
            const void* src = nullptr;
            if (rc == AM_OK && d_parts) {
                src = d_parts[i];
                if (!src) rc = fail(AM_ERR_INVALID_ARG, "null part pointer");
                else rc = check_resident(src, sl.device, i);
            } else if (rc == AM_OK) {
                if (pt.n * 4 > sl.ring_cap) {   // (the ring of the host-buffer batch path: its first half holds the part)
                    for (void*& r : sl.ring) { if (r) (void)hipFree(r); r = nullptr; }
                    sl.ring_cap = 0;
                    for (void*& r : sl.ring) {
                        const hipError_t e = hipMalloc(&r, pt.n * 4);
                        if (e != hipSuccess) { r = nullptr; rc = hip_fail(e, "hipMalloc(pool ring)"); break; }
                    }
                    if (rc == AM_OK) sl.ring_cap = pt.n * 4;
                }
                if (rc == AM_OK) {
                    hipError_t e = hipMemcpyAsync(sl.ring[0], static_cast<const char*>(host_hay) + 4 * pt.a, pt.n * 4, hipMemcpyHostToDevice,
                                                  sl.copy_stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(sl.copy_stream);
                    if (e != hipSuccess) rc = hip_fail(e, "host-to-device copy (long haystack)");
                    src = sl.ring[0];
                }
            }
            if (rc == AM_OK) {
                am_needle* h = sl.needle;
                std::lock_guard<std::recursive_mutex> lk2(h->ctx->mu);
                PartSpec spec{pt.nw, (uint64_t)pt.a, pt.w0, total_windows, &raw[i]};
                size_t n = 0;
                rc = match_many(h, &src, &pt.n, 1, p, nullptr, 0, &n, fmt, 0, 1, true, nullptr, &spec);
            }
            rcs[i] = rc;
            if (rc) errs[i] = t_err;
        });
    for (std::thread& th : threads) th.join();
    for (size_t i = 0; i < nslots; ++i)
        if (rcs[i]) { t_err = errs[i]; return rcs[i]; }
    // flatten in window order, then ONE sort + overshadow pass over the union (audio_matcher.rs:132-140): a peak
    // next to a cut sees its neighbour from the other part, exactly as in a single call
    std::vector<am_peak> all;
    for (size_t i = 0; i < nslots; ++i) all.insert(all.end(), raw[i].begin(), raw[i].end());
    const int rc = merge_peaks(all, p, snapshot_opts(nullptr).surrounding_from != 0, out, cap, n_out);
    if (hooks.fn) hooks.fn(hooks.user, 0, 1, total_windows);
    return rc;
}

}  // namespace

int am_pool_match_long(am_pool* pool, const void* haystack, size_t len, int sample_format, const am_match_params* p,
                       am_peak* out, size_t cap, size_t* n_out) {
    if (!haystack && len) return fail(AM_ERR_INVALID_ARG, "null pointer");
    return pool_long(pool, haystack, nullptr, len, sample_format, p, out, cap, n_out);
}

int am_pool_match_long_device(am_pool* pool, const void* const* d_parts, size_t len, int sample_format, const am_match_params* p,
                              am_peak* out, size_t cap, size_t* n_out) {
    if (!d_parts) return fail(AM_ERR_INVALID_ARG, "null pointer");
    return pool_long(pool, nullptr, d_parts, len, sample_format, p, out, cap, n_out);
}

int am_pool_needle_count(const am_pool* pool, size_t* n_needles) {
    if (!pool || !n_needles) return fail(AM_ERR_INVALID_ARG, "null pointer");
    *n_needles = pool->slots.empty() ? 0 : pool->slots[0].needles.size();
    return AM_OK;
}

int am_set_progress_callback(am_progress_fn fn, void* user) {
    std::lock_guard<std::mutex> lk(g_hooks_mu);
    g_hooks.fn = fn;
    g_hooks.user = user;
    return AM_OK;
}

int am_set_chunk_progress_callback(am_chunk_progress_fn fn, void* user) {
    std::lock_guard<std::mutex> lk(g_hooks_mu);
    g_hooks.chunk_fn = fn;
    g_hooks.chunk_user = user;
    return AM_OK;
}

// Measurement hook (not part of the drop-in boundary): the column kernels K1 and K3 on `npairs` block pairs of
// synthetic input, `iters` launches each, average launch time in ms.  wide = 0: the production 2^22 plan
// (512 x 8192); wide = 1: 2^23 factored 512 x 16384 -- the same 512-row kernels on rows twice as long (no row kernel
// exists for that plan yet: this sizes what it would be worth).  dense: K3 writes every run / none.
int am_debug_column_bench(int device, int wide, int npairs, int iters, int dense, double* k1_ms, double* k3_ms) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (npairs < 1 || npairs > 64 || iters < 1 || !k1_ms || !k3_ms) return fail(AM_ERR_INVALID_ARG, "bad arguments");
    const Plan* pl = nullptr;
    if ((rc = wide ? get_plan(c, 23, &pl, 9) : get_plan(c, 22, &pl))) return rc;
    const long long N = 1ll << pl->dev.logN, s = 441000, hop = ((N - s + 1) / kTile) * kTile;
    const long long nblocks = 2ll * npairs, out_count = nblocks * hop, src_len = out_count + s - 1 + kTile;
    DevBuf src, work, scores, stats32, side;
    struct Release { DevBuf* b[5]; ~Release() { for (DevBuf* x : b) x->release(); } } release_all{{&src, &work, &scores, &stats32, &side}};
    if ((rc = src.ensure((size_t)src_len * 4)) || (rc = work.ensure((size_t)npairs * (size_t)N * sizeof(float2))) ||
        (rc = scores.ensure((size_t)out_count * 4)) || (rc = stats32.ensure((size_t)(out_count / 32) * sizeof(float2))) ||
        (rc = side.ensure(sparse_bytes(nblocks, pl->dev)))) return rc;
    AM_HIP(launch_synth(c->stream, (float*)src.p, 7, 1, 0, src_len, 0.25f));
    Job job{};
    job.src = src.p; job.src_len = src_len; job.lead = 0; job.src_kind = 0;
    job.dst = (float*)scores.p; job.out_count = out_count; job.hop = (int)hop; job.nblocks = (int)nblocks; job.first_pair = 0;
    ScanCfg scan{};
    fill_scan_cfg(&scan, stats32.p, side.p, nblocks, pl->dev, dense ? -1.0f : 1e30f, FLT_MAX, 60 * 44100, 70 * 44100 - s);
    hipEvent_t e[3];
    for (auto& x : e) AM_HIP(hipEventCreate(&x));
    double t1 = 0, t3 = 0;
    for (int it = -2; it < iters; ++it) {   // (two untimed rounds first)
        AM_HIP(hipEventRecord(e[0], c->stream));
        AM_HIP(launch_k1(c->stream, job, npairs, (float2*)work.p, pl->dev, 0));
        AM_HIP(hipEventRecord(e[1], c->stream));
        AM_HIP(launch_k3(c->stream, job, npairs, (const float2*)work.p, pl->dev, 1e-3f, scan, 0, false));
        AM_HIP(hipEventRecord(e[2], c->stream));
        AM_HIP(hipStreamSynchronize(c->stream));
        float a = 0, b = 0;
        AM_HIP(hipEventElapsedTime(&a, e[0], e[1]));
        AM_HIP(hipEventElapsedTime(&b, e[1], e[2]));
        if (it >= 0) { t1 += a; t3 += b; }
    }
    for (auto& x : e) (void)hipEventDestroy(x);
    *k1_ms = t1 / iters; *k3_ms = t3 / iters;
    return AM_OK;
}

int am_profile_enable(int device, int on) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    prof_harvest(c);
    c->prof = on != 0;
    return AM_OK;
}
int am_profile_reset(int device) {
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    prof_harvest(c);
    for (int i = 0; i < KN_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    return AM_OK;
}
int am_profile_query(int device, const char* kernel, double* total_ms, uint64_t* launches) {
    if (!kernel || !total_ms || !launches) return fail(AM_ERR_INVALID_ARG, "null pointer");
    Ctx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    prof_harvest(c);
    double ms = 0; uint64_t n = 0; bool found = false;
    for (int i = 0; i < KN_COUNT; ++i) {
        if (!strcmp(kernel, "*") || !strcmp(kernel, kKernelNames[i])) { ms += c->prof_ms[i]; n += c->prof_n[i]; found = true; }
    }
    if (!found) return fail(AM_ERR_INVALID_ARG, "unknown kernel name");
    *total_ms = ms; *launches = n;
    return AM_OK;
}

int am_set_option(const char* key, long long value) {
    if (!key) return fail(AM_ERR_INVALID_ARG, "null key");
    if (!strcmp(key, "log_n")) {
        if (value != 0 && (value < kLogNMin || value > kLogNMax)) return fail(AM_ERR_INVALID_ARG, "log_n out of range");
        g_opt_log_n = value; return AM_OK;
    }
    if (!strcmp(key, "half_pipeline")) { g_opt_half = value <= 0 ? 0 : (value >= 2 ? 2 : 1); return AM_OK; }
    if (!strcmp(key, "batch_overlap")) { g_opt_batch_overlap = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "profile_every")) { g_opt_profile_every = value < 1 ? 1 : value; return AM_OK; }
    if (!strcmp(key, "dense_scores")) { g_opt_dense = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "tail_block")) { g_opt_tail_block = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "host_pick_wait")) { g_opt_host_pick_wait = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "device_redo")) { g_opt_device_redo = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "k3_group")) { g_opt_k3_group = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "pick_group")) { g_opt_pick_group = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "k2_mfma")) { set_k2_mfma(value != 0); return AM_OK; }
    if (!strcmp(key, "pick_stream_priority")) { g_opt_pick_priority = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "peak_filter_order")) { g_opt_peak_filter_order = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "distance_rule")) {
        if (value < 0 || value > 3) return fail(AM_ERR_INVALID_ARG, "distance_rule out of range (bit 0: inclusive, bit 1: between plateau starts)");
        g_opt_distance_rule = value; return AM_OK;
    }
    if (!strcmp(key, "tail_window")) { g_opt_tail_window = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "surrounding_from")) { g_opt_surrounding_from = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "debug_no_realloc")) { g_opt_debug_no_realloc = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "debug_redo_arm_at")) { g_opt_debug_redo_arm_at = value < -1 ? -2 : value; return AM_OK; }
    if (!strcmp(key, "needle_group")) {
        if (value < 1 || value > kMaxNeedleGroup) return fail(AM_ERR_INVALID_ARG, "needle_group out of range");
        g_opt_needle_group = value; return AM_OK;
    }
    if (!strcmp(key, "profile_mask")) { g_opt_profile_mask = value; return AM_OK; }
    if (!strcmp(key, "pairs_per_group")) {
        if (value < 1 || value > 64) return fail(AM_ERR_INVALID_ARG, "pairs_per_group out of range");
        g_opt_pairs_per_group = value; return AM_OK;
    }
    return fail(AM_ERR_INVALID_ARG, "unknown option");
}
int am_get_option(const char* key, long long* value) {
    if (!key || !value) return fail(AM_ERR_INVALID_ARG, "null pointer");
    if (!strcmp(key, "log_n")) { *value = g_opt_log_n; return AM_OK; }
    if (!strcmp(key, "pairs_per_group")) { *value = g_opt_pairs_per_group; return AM_OK; }
    if (!strcmp(key, "half_pipeline")) { *value = g_opt_half; return AM_OK; }
    if (!strcmp(key, "needle_group")) { *value = g_opt_needle_group; return AM_OK; }
    if (!strcmp(key, "batch_overlap")) { *value = g_opt_batch_overlap; return AM_OK; }
    if (!strcmp(key, "profile_every")) { *value = g_opt_profile_every; return AM_OK; }
    if (!strcmp(key, "dense_scores")) { *value = g_opt_dense; return AM_OK; }
    if (!strcmp(key, "tail_block")) { *value = g_opt_tail_block; return AM_OK; }
    if (!strcmp(key, "host_pick_wait")) { *value = g_opt_host_pick_wait; return AM_OK; }
    if (!strcmp(key, "device_redo")) { *value = g_opt_device_redo; return AM_OK; }
    if (!strcmp(key, "k3_group")) { *value = g_opt_k3_group; return AM_OK; }
    if (!strcmp(key, "pick_group")) { *value = g_opt_pick_group; return AM_OK; }
    if (!strcmp(key, "k2_mfma")) { *value = k2_mfma_enabled() ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "pick_stream_priority")) { *value = g_opt_pick_priority; return AM_OK; }
    if (!strcmp(key, "peak_filter_order")) { *value = g_opt_peak_filter_order; return AM_OK; }
    if (!strcmp(key, "distance_rule")) { *value = g_opt_distance_rule; return AM_OK; }
    if (!strcmp(key, "tail_window")) { *value = g_opt_tail_window; return AM_OK; }
    if (!strcmp(key, "surrounding_from")) { *value = g_opt_surrounding_from; return AM_OK; }
    if (!strcmp(key, "debug_no_realloc")) { *value = g_opt_debug_no_realloc; return AM_OK; }
    if (!strcmp(key, "debug_redo_arm_at")) { *value = g_opt_debug_redo_arm_at; return AM_OK; }
    if (!strcmp(key, "profile_mask")) { *value = g_opt_profile_mask; return AM_OK; }
    return fail(AM_ERR_INVALID_ARG, "unknown option");
}

int am_needle_set_option(am_needle* h, const char* key, long long value) {
    if (!h || !h->ctx || !key) return fail(AM_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    if (!strcmp(key, "log_n")) {
        if (value > 0 && (value < kLogNMin || value > kLogNMax)) return fail(AM_ERR_INVALID_ARG, "log_n out of range");
        h->opt_log_n = value < 0 ? -1 : value; return AM_OK;
    }
    if (!strcmp(key, "half_pipeline")) { h->opt_half = value < 0 ? -1 : (value >= 2 ? 2 : (value ? 1 : 0)); return AM_OK; }
    return fail(AM_ERR_INVALID_ARG, "unknown per-handle option (log_n, half_pipeline)");
}
int am_needle_get_option(const am_needle* h, const char* key, long long* value) {
    if (!h || !h->ctx || !key || !value) return fail(AM_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::recursive_mutex> lk(h->ctx->mu);
    if (!strcmp(key, "log_n")) { *value = h->opt_log_n; return AM_OK; }
    if (!strcmp(key, "half_pipeline")) { *value = h->opt_half; return AM_OK; }
    return fail(AM_ERR_INVALID_ARG, "unknown per-handle option (log_n, half_pipeline)");
}

}  // extern "C"
