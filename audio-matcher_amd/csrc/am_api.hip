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
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
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

// progress hook (audio_matcher.rs:102-117, 129)
static am_progress_fn g_progress_fn = nullptr;
static void* g_progress_user = nullptr;

// tuning knobs
static long long g_opt_log_n = 0;          // 0 = auto
static long long g_opt_pairs_per_group = 64;
static long long g_opt_profile_mask = -1;    // bit i = bracket kernel class i with events while profiling is on
static long long g_opt_lanes = 1;           // 2 = overlap the kernels of alternate pair groups on two streams
static long long g_opt_half = 0;            // 1 = half-precision storage of the work matrix (config 5)
static long long g_opt_batch_overlap = 1;   // 1 = in a batch, pick the peaks of haystack k beside the transforms of k+1
static long long g_opt_needle_group = 8;    // needles sharing one forward row transform in am_match_multi_device
static const float kHalfGain = 1024.0f;      // keeps the stored values of a normalised score near 1
static const double kMinEfficiency = 0.75;  // hop / N the auto plan accepts
static const int kLogNMin = 10, kLogNMax = 23;

// ---------------------------------------------------------------------------
static long long g_opt_vmm = 0;   // experiment: back the work matrix with hipMemCreate chunks (see DevBuf)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // EXPERIMENT (option "vmm_work"): virtual-memory-management allocation, one physical
    // handle per `chunk` bytes, to see whether the physical fragment size behind the work
    // matrix explains the two speeds K1/K3 show from process to process.
    bool vmm = false;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t vmm_chunk = 0;
    int ensure_vmm(size_t bytes, int device) {
        release();
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        AM_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        const size_t chunk = std::max<size_t>(gran, (size_t)g_opt_vmm << 20);
        const size_t total = (bytes + chunk - 1) / chunk * chunk;
        AM_HIP(hipMemAddressReserve(&p, total, chunk, nullptr, 0));
        for (size_t off = 0; off < total; off += chunk) {
            hipMemGenericAllocationHandle_t hnd;
            AM_HIP(hipMemCreate(&hnd, chunk, &prop, 0));
            handles.push_back(hnd);
            AM_HIP(hipMemMap(static_cast<char*>(p) + off, chunk, 0, hnd, 0));
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        AM_HIP(hipMemSetAccess(p, total, &acc, 1));
        vmm = true; vmm_chunk = chunk; cap = total;
        return AM_OK;
    }
    int ensure(size_t bytes, int vmm_device = -1) {
        if (bytes <= cap) return AM_OK;
        if (vmm_device >= 0 && g_opt_vmm > 0) return ensure_vmm(bytes + bytes / 8, vmm_device);
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
        if (p && vmm) {
            (void)hipDeviceSynchronize();
            (void)hipMemUnmap(p, cap);
            for (auto hnd : handles) (void)hipMemRelease(hnd);
            (void)hipMemAddressFree(p, cap);
            handles.clear();
        } else if (p) {
            (void)hipFree(p);
        }
        p = nullptr; cap = 0; vmm = false;
    }
};
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    unsigned flags = hipHostMallocDefault;
    int ensure(size_t bytes) {
        if (bytes <= cap) return AM_OK;
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
};

struct ProfRec { int name; hipEvent_t e0, e1; };
static const char* kKernelNames[] = {"k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks", "other"};
enum { KN_K1 = 0, KN_K2, KN_K3, KN_STATS, KN_PEAKS, KN_OTHER, KN_COUNT };

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // second lane of the two-lane block pipeline
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    std::recursive_mutex mu;
    std::map<int, Plan> plans;
    DevBuf work, work2, scores, stats, stats32, wflags, segs, peaks, io_in, io_out, sum;
    // second set of the score-side buffers: in a batch the peak pick of haystack k runs on
    // stream2 beside the transforms of haystack k+1, which then need their own set
    DevBuf scores_b, stats_b, stats32_b, wflags_b, peaks_b;
    hipEvent_t ev_k3[2] = {nullptr, nullptr}, ev_pick[2] = {nullptr, nullptr};
    HostBuf pinned;
    // Per-chunk result headers live in coherent pinned host memory that the peak
    // kernel writes directly (a few KB per haystack): no device-to-host copy
    // sits between the last kernel and the host's wake-up.
    HostBuf hdr;
    // the chunk list currently resident in `segs` (re-uploaded only when it changes)
    std::vector<Segment> segs_resident;
    // profiling
    bool prof = false;
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    double prof_ms[KN_COUNT] = {0};
    uint64_t prof_n[KN_COUNT] = {0};
};

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
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return hip_fail(se, "hipStreamCreate"); }
    (void)hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&c->ev_c, hipEventDisableTiming);
    for (int i = 0; i < 2; ++i) {
        (void)hipEventCreateWithFlags(&c->ev_k3[i], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&c->ev_pick[i], hipEventDisableTiming);
    }
    g_ctx[device] = c;
    *out = c;
    return AM_OK;
}

// ---- profiling helpers ------------------------------------------------------
static hipEvent_t prof_event(Ctx* c) {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    Ctx* c; int name; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on;
    ProfScope(Ctx* c_, int name_, hipStream_t st_ = nullptr) : c(c_), name(name_), st(st_ ? st_ : c_->stream) {
        on = c->prof && ((g_opt_profile_mask >> name) & 1);
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

static int get_plan(Ctx* c, int logN, const Plan** out) {
    auto it = c->plans.find(logN);
    if (it != c->plans.end()) { *out = &it->second; return AM_OK; }
    if (logN < kLogNMin || logN > kLogNMax) return fail(AM_ERR_INVALID_ARG, "unsupported transform size");
    Plan p;
    int logN1 = logN - 13;
    if (logN1 < kColsLog) logN1 = kColsLog;
    if (logN1 > 9) logN1 = 9;
    int logN2 = logN - logN1;
    // N = 2^22 runs on the register kernels as 256 columns x two 8192-point row halves
    const int wide = logN == 22 ? 1 : 0;
    if (wide) { logN1 = 8; logN2 = 13; }
    const int logLo = (logN + 1) / 2;
    const size_t n1h = (size_t)1 << (logN1 - 1), n2h = (size_t)1 << (logN2 - 1);
    const size_t nlo = (size_t)1 << logLo, nhi = (size_t)1 << (logN - logLo);
    std::vector<float2> host(n1h + n2h + nlo + nhi);
    fill_twiddles(host, 0, n1h, (double)(1u << logN1), 1.0);
    fill_twiddles(host, n1h, n2h, (double)(1u << logN2), 1.0);
    fill_twiddles(host, n1h + n2h, nlo, (double)((size_t)1 << logN), 1.0);
    fill_twiddles(host, n1h + n2h + nlo, nhi, (double)((size_t)1 << logN), (double)nlo);
    AM_HIP(hipMalloc((void**)&p.tables, host.size() * sizeof(float2)));
    AM_HIP(copy_on_stream(c, p.tables, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice));
    p.dev.logN = logN; p.dev.logN1 = logN1; p.dev.logN2 = logN2; p.dev.logLo = logLo; p.dev.wide = wide;
    p.dev.tw1 = p.tables;
    p.dev.tw2 = p.tables + n1h;
    p.dev.twlo = p.tables + n1h + n2h;
    p.dev.twhi = p.tables + n1h + n2h + nlo;
    auto ins = c->plans.emplace(logN, p);
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
    // lowest chunk minimum of the scaled scores seen so far (per scale mode);
    // drives the raw-score write threshold of the fused scan
    bool have_min[2] = {false, false};
    float min_seg_min[2] = {0.f, 0.f};
};

namespace am {

static int pick_log_n(size_t s, long long out_count, int* logN_out) {
    // smallest transform that can hold the needle at all
    int min_log = kLogNMin;
    while (min_log <= kLogNMax && ((size_t)1 << min_log) < s + 1) ++min_log;
    if (min_log > kLogNMax) return fail(AM_ERR_INVALID_ARG, "needle too long for the largest transform (2^23)");
    if (g_opt_log_n > 0) {
        int l = (int)g_opt_log_n;
        if (l < min_log) l = min_log;
        if (l > kLogNMax) l = kLogNMax;
        *logN_out = l;
        return AM_OK;
    }
    const long long span = out_count + (long long)s - 1;
    // The register-resident kernels exist for N = 2^21 only and are several times
    // faster per point than the generic ones, so every problem that is not small
    // runs on N = 2^21 as long as at least a quarter of each block is new output
    // (needles up to ~1.5 M samples); short needles simply get a longer hop.
    if (span > (1ll << 19)) {
        // N = 2^21 (256 x 8192) while at least 3/4 of a block is new output; N = 2^22
        // (256 x 2 x 8192, about 8 % dearer per point) for longer needles, up to 3 M
        // samples; beyond that the generic kernels at N = 2^23
        // measured crossover (tools/needle_sweep.py): 2^22 wins from about 10.5 s of 44.1 kHz audio
        // (the half-precision work matrix exists on the 2^21 plan only: keep it there up to 2^19 samples)
        if ((long long)s <= (g_opt_half ? (1ll << 19) : 7 * (1ll << 16))) { *logN_out = 21; return AM_OK; }
        if ((long long)s <= (1ll << 22) - (1ll << 20)) { *logN_out = 22; return AM_OK; }
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
    // the two K2 forms keep the spectrum in different (register-order) layouts
    const int key = pl->dev.logN * 4 + (plan_k2_is_r16(pl->dev) ? 1 + g_k2_variant : 0);
    auto it = h->spectra.find(key);
    if (it != h->spectra.end()) { *out = it->second; return AM_OK; }
    const size_t N = (size_t)1 << pl->dev.logN;
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

// The overlap-save engine: scores[j] = factor * sum_n X[j + n - lead] needle[n]
// When want_stats is set and the plan supports it, K3 also writes the level-0
// (min,max) summary into c->stats32 and *have_stats becomes true.
struct ScanRequest {
    float theta;             // in: raw-score write threshold
    long long seg_c, seg_d;  // in: chunk geometry (scores i*seg_c .. i*seg_c + seg_d)
    int set;                 // in: which set of score-side buffers (0, or 1 in an overlapped batch)
    hipEvent_t before_k3;    // in: K3 must not overwrite that set before this event (or null)
    bool fused;              // out: K3 produced stats32 / wflags
    SparseScores sparse;     // out: description of what was written
};
static int run_correlation(am_needle* h, const void* d_src, long long src_len, long long lead,
                           float* d_dst, long long out_count, float factor,
                           ScanRequest* scan_req = nullptr, int src_kind = 0) {
    Ctx* c = h->ctx;
    int logN = 0;
    int rc = pick_log_n(h->n, out_count, &logN);
    if (rc) return rc;
    const Plan* pl = nullptr;
    if ((rc = get_plan(c, logN, &pl))) return rc;
    const float2* hc = nullptr;
    if ((rc = needle_spectrum(h, pl, &hc))) return rc;
    const long long N = 1ll << logN;
    long long hop = N - (long long)h->n + 1;
    if (hop >= 8 * kTile) hop = (hop / kTile) * kTile;
    const long long nblocks = (out_count + hop - 1) / hop;
    const long long npairs = (nblocks + 1) / 2;
    long long ppg = std::max<long long>(1, g_opt_pairs_per_group);
    if (ppg > npairs) ppg = npairs;
    if ((rc = c->work.ensure((size_t)ppg * (size_t)N * sizeof(float2), c->device))) return rc;
    ScanCfg scan{};
    if (scan_req) {
        scan_req->fused = false;
        scan_req->sparse = SparseScores{nullptr, nullptr, 0.f, (int)hop, pl->dev.logN2 + pl->dev.wide, 1.0 / (double)hop};
        if (plan_is_r16(pl->dev) && (hop % kTile) == 0) {
            DevBuf& b32 = scan_req->set ? c->stats32_b : c->stats32;
            DevBuf& bwf = scan_req->set ? c->wflags_b : c->wflags;
            if ((rc = b32.ensure((size_t)((out_count + 31) / 32) * sizeof(float2)))) return rc;
            if ((rc = bwf.ensure((size_t)nblocks << (pl->dev.logN2 + pl->dev.wide - kColsLog)))) return rc;
            scan.stats32 = (float2*)b32.p;
            scan.wflags = (unsigned char*)bwf.p;
            scan.theta = scan_req->theta;
            scan.seg_c = scan_req->seg_c;
            scan.seg_d = scan_req->seg_d;
            scan.inv_c = scan.seg_c > 0 ? 1.0 / (double)scan.seg_c : 0.0;
            scan_req->fused = true;
            scan_req->sparse = SparseScores{scan.wflags, scan.stats32, scan.theta, (int)hop, pl->dev.logN2 + pl->dev.wide, 1.0 / (double)hop};
        }
    }
    // half-precision storage of the work matrix: K2 normalises by the needle
    // energy (times a fixed gain) so that stored values sit mid-range in f16
    const bool half = g_opt_half && plan_is_r16(pl->dev) && !pl->dev.wide && g_k2_variant == 0;
    const float hscale = half ? kHalfGain * h->inv_autocorr : 1.0f;
    const float k3scale = half ? factor / hscale : factor;
    Job job{};
    job.src = d_src; job.src_len = src_len; job.lead = lead; job.src_kind = src_kind;
    job.dst = d_dst; job.out_count = out_count; job.hop = (int)hop; job.nblocks = (int)nblocks;
    const bool two_lanes = g_opt_lanes == 2 && c->stream2 && c->ev_a && c->ev_b && c->ev_c && npairs >= 8 && ppg >= npairs;
    if (!two_lanes) {
        for (long long first = 0; first < npairs; first += ppg) {
            const int np = (int)std::min(ppg, npairs - first);
            job.first_pair = (int)first;
            { ProfScope ps(c, KN_K1); AM_HIP(launch_k1(c->stream, job, np, (float2*)c->work.p, pl->dev, half)); }
            { ProfScope ps(c, KN_K2); AM_HIP(launch_k2(c->stream, np, (float2*)c->work.p, hc, pl->dev, nullptr, half, hscale)); }
            if (first == 0 && scan_req && scan_req->before_k3) AM_HIP(hipStreamWaitEvent(c->stream, scan_req->before_k3, 0));
            { ProfScope ps(c, KN_K3); AM_HIP(launch_k3(c->stream, job, np, (const float2*)c->work.p, pl->dev, k3scale, scan, half)); }
        }
        return AM_OK;
    }
    // Two-lane pipeline: the pairs are cut into four groups that alternate between
    // two streams, the second lane one kernel behind the first, so that a
    // bandwidth-heavy kernel of one group runs beside a latency/VALU-heavy kernel
    // of the other.  Groups write disjoint parts of the score arrays; each lane has
    // its own half of the work matrix (the whole matrix was sized for all pairs).
    const long long gsz = (npairs + 3) / 4;
    hipStream_t lane[2] = {c->stream, c->stream2};
    // each lane owns gsz pair slots of the work matrix; groups g and g+2 reuse them in stream order
    float2* wk[2] = {(float2*)c->work.p, (float2*)c->work.p + (size_t)gsz * (size_t)N};
    AM_HIP(hipEventRecord(c->ev_a, c->stream));            // everything queued before this call
    AM_HIP(hipStreamWaitEvent(c->stream2, c->ev_a, 0));
    int g = 0;
    for (long long first = 0; first < npairs; first += gsz, ++g) {
        const int ln = g & 1;
        const int np = (int)std::min(gsz, npairs - first);
        job.first_pair = (int)first;
        { ProfScope ps(c, KN_K1, lane[ln]); AM_HIP(launch_k1(lane[ln], job, np, wk[ln], pl->dev, half)); }
        if (g == 0) {                                       // stagger: lane 1 starts after lane 0's first K1
            AM_HIP(hipEventRecord(c->ev_b, lane[0]));
            AM_HIP(hipStreamWaitEvent(lane[1], c->ev_b, 0));
        }
        { ProfScope ps(c, KN_K2, lane[ln]); AM_HIP(launch_k2(lane[ln], np, wk[ln], hc, pl->dev, nullptr, half, hscale)); }
        { ProfScope ps(c, KN_K3, lane[ln]); AM_HIP(launch_k3(lane[ln], job, np, wk[ln], pl->dev, k3scale, scan, half)); }
    }
    AM_HIP(hipEventRecord(c->ev_c, c->stream2));           // join: the main stream continues after both lanes
    AM_HIP(hipStreamWaitEvent(c->stream, c->ev_c, 0));
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

// Launches find_peaks (audio_matcher.rs:221-230) for `nsegs` segments of a
// resident score array; segment descriptors and result headers live at
// [seg_off, seg_off + nsegs) of the context's segment / header buffers.
static int launch_pick(Ctx* c, const float* d_scores, long long n_scores, int seg_off, int nsegs,
                       float min_prom, long long min_dist, const ScanRequest* scan, int hdr_off = -1,
                       hipStream_t st = nullptr) {
    if (hdr_off < 0) hdr_off = seg_off;
    if (!st) st = c->stream;
    const int set = scan ? scan->set : 0;
    DevBuf& bstats = set ? c->stats_b : c->stats;
    DevBuf& bpeaks = set ? c->peaks_b : c->peaks;
    const float2* d_stats32 = (scan && scan->fused) ? scan->sparse.stats32 : nullptr;
    const SparseScores sp = (scan && scan->fused) ? scan->sparse : SparseScores{nullptr, nullptr, 0.f, 1, 5, 1.0};
    if (nsegs == 0 || n_scores <= 0) return AM_OK;
    int rc;
    const long long ntiles = (n_scores + kTile - 1) / kTile;
    if ((rc = bstats.ensure((size_t)ntiles * sizeof(float2)))) return rc;
    {
        ProfScope ps(c, KN_STATS, st);
        if (d_stats32) AM_HIP(launch_stats_reduce(st, d_stats32, n_scores, (float2*)bstats.p));
        else AM_HIP(launch_tile_stats(st, d_scores, n_scores, (float2*)bstats.p));
    }
    {
        ProfScope ps(c, KN_PEAKS, st);
        AM_HIP(launch_peaks(st, d_scores, n_scores, (const float2*)bstats.p,
                            (const Segment*)c->segs.p + seg_off, nsegs, min_prom, min_dist,
                            (am_peak*)bpeaks.p, (SegHeader*)c->hdr.p + hdr_off, sp));
    }
    return AM_OK;
}

// windows of common::chunked(chunk + overlap, hop = chunk) (audio_matcher.rs:104)
// as slices of the global score array; a window shorter than the needle has
// no valid lag and is skipped.
static void make_segments(size_t len, size_t s, const am_match_params* p, std::vector<Segment>& segs) {
    const unsigned long long window = p->chunk + p->overlap;
    for (unsigned long long off = 0; off < len; off += p->chunk) {
        const unsigned long long w = std::min<unsigned long long>(window, len - off);
        if (w < s) continue;
        Segment sg; sg.a = (long long)off; sg.b = (long long)(off + w - s + 1);
        segs.push_back(sg);
    }
}

// sort by start (audio_matcher.rs:135) + filter_surrounding (audio_matcher.rs:136-139)
static int merge_peaks(std::vector<am_peak>& all, const am_match_params* p, am_peak* out, size_t cap, size_t* n_out) {
    std::stable_sort(all.begin(), all.end(), [](const am_peak& x, const am_peak& y) { return x.start < y.start; });
    size_t n = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        const am_peak* before = i > 0 ? &all[i - 1] : nullptr;
        const am_peak* after = i + 1 < all.size() ? &all[i + 1] : nullptr;
        if (is_overshadowed(all[i], before, p->sr, p->overshadow_distance_s) ||
            is_overshadowed(all[i], after, p->sr, p->overshadow_distance_s))
            continue;
        if (n < cap) out[n] = all[i];
        ++n;
    }
    *n_out = n;
    if (n > cap) return fail(AM_ERR_CAPACITY, "peak output buffer too small");
    return AM_OK;
}

// calc_chunks (audio_matcher.rs:88-141) over a batch of resident haystacks =
// the per-file loop of matcher::run (matcher/mod.rs:42-87).  Everything is
// queued on the context's stream without host synchronisation; one small
// device-to-host copy of the per-chunk headers ends the batch.
static int match_many(am_needle* h, const void* const* d_hays, const size_t* lens, size_t n_hay,
                      const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out, int src_kind = 0) {
    Ctx* c = h->ctx;
    const size_t s = h->n;
    if (p->chunk == 0) return fail(AM_ERR_INVALID_ARG, "chunk must be > 0");
    if (p->scale != AM_SCALE_NONE && p->scale != AM_SCALE_LIB)
        return fail(AM_ERR_INVALID_ARG, "am_match supports AM_SCALE_NONE and AM_SCALE_LIB (AM_SCALE_MY depends on the window length)");
    const float factor = scale_factor(h, p->scale, 1);
    // Raw scores are written only where some score >= theta.  theta sits half a
    // prominence above the lowest chunk minimum seen with this needle; the first
    // call (no history) writes everything.  The peak kernel certifies per chunk
    // that theta was low enough; a failed certificate redoes that haystack.
    const int sm = p->scale == AM_SCALE_LIB ? 1 : 0;
    ScanRequest scan{};
    scan.theta = (h->have_min[sm] && p->min_prominence > 0.f) ? h->min_seg_min[sm] + 0.5f * p->min_prominence : -FLT_MAX;
    scan.seg_c = (long long)p->chunk;
    scan.seg_d = (long long)(p->chunk + p->overlap) - (long long)s;
    std::vector<Segment> segs;
    std::vector<int> seg_off(n_hay + 1, 0);
    size_t max_scores = 0, max_segs = 0;
    for (size_t k = 0; k < n_hay; ++k) {
        n_out[k] = 0;
        seg_off[k] = (int)segs.size();
        if (d_hays[k] && lens[k] >= s) {
            make_segments(lens[k], s, p, segs);
            max_scores = std::max(max_scores, lens[k] - s + 1);
        }
        max_segs = std::max(max_segs, segs.size() - (size_t)seg_off[k]);
    }
    seg_off[n_hay] = (int)segs.size();
    const size_t nsegs = segs.size();
    if (nsegs == 0) return AM_OK;
    if (max_segs > (size_t)1 << 18 || nsegs > (size_t)1 << 24)
        return fail(AM_ERR_INVALID_ARG, "chunk size too small for this haystack (more than 2^18 chunks)");
    int rc;
    const size_t hdr_bytes = sizeof(SegHeader) * nsegs;
    // In a batch the peak pick of haystack k (small, latency-bound kernels) runs on a second
    // stream beside the transforms of haystack k+1; the score-side buffers alternate between
    // two sets and K3 waits for the pick that last read the set it is about to overwrite.
    size_t n_active = 0;
    for (size_t k = 0; k < n_hay; ++k) n_active += seg_off[k + 1] > seg_off[k];
    const bool overlap = g_opt_batch_overlap && n_active > 1 && g_opt_lanes != 2 && c->stream2 &&
                         c->ev_k3[0] && c->ev_k3[1] && c->ev_pick[0] && c->ev_pick[1];
    if ((rc = c->scores.ensure(max_scores * sizeof(float)))) return rc;
    if ((rc = c->hdr.ensure(hdr_bytes))) return rc;
    if ((rc = c->peaks.ensure(sizeof(am_peak) * max_segs * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    if (overlap) {
        if ((rc = c->scores_b.ensure(max_scores * sizeof(float)))) return rc;
        if ((rc = c->peaks_b.ensure(sizeof(am_peak) * max_segs * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    }
    if ((rc = upload_segments(c, segs))) return rc;
    SegHeader* h_hdr = static_cast<SegHeader*>(c->hdr.p);
    size_t seq = 0;
    for (size_t k = 0; k < n_hay; ++k) {
        const int ns = seg_off[k + 1] - seg_off[k];
        if (ns == 0) continue;
        if (g_progress_fn) g_progress_fn(g_progress_user, k, 0, (size_t)ns);
        const long long out_count = (long long)(lens[k] - s + 1);
        const int set = overlap ? (int)(seq & 1) : 0;
        float* d_scores = (float*)(set ? c->scores_b.p : c->scores.p);
        scan.set = set;
        scan.before_k3 = (overlap && seq >= 2) ? c->ev_pick[set] : nullptr;
        if ((rc = run_correlation(h, d_hays[k], (long long)lens[k], 0, d_scores, out_count, factor,
                                  &scan, src_kind))) return rc;
        if (overlap) {
            AM_HIP(hipEventRecord(c->ev_k3[set], c->stream));
            AM_HIP(hipStreamWaitEvent(c->stream2, c->ev_k3[set], 0));
        }
        if ((rc = launch_pick(c, d_scores, out_count, seg_off[k], ns, p->min_prominence,
                              (long long)p->min_distance, &scan, -1, overlap ? c->stream2 : c->stream))) return rc;
        if (overlap) AM_HIP(hipEventRecord(c->ev_pick[set], c->stream2));
        ++seq;
    }
    AM_HIP(hipStreamSynchronize(c->stream));   // the headers are in host memory once the peak kernels have finished
    if (overlap) AM_HIP(hipStreamSynchronize(c->stream2));
    scan.set = 0;
    scan.before_k3 = nullptr;
    int worst = AM_OK;
    std::vector<am_peak> all;
    for (size_t k = 0; k < n_hay; ++k) {
        const int s0 = seg_off[k], s1 = seg_off[k + 1];
        if (s1 == s0) continue;
        bool big = false;
        for (int i = s0; i < s1; ++i) {
            if (h_hdr[i].overflow & 1) return fail(AM_ERR_PEAK_OVERFLOW, "more than AM_MAX_PEAKS_PER_CHUNK peaks in one chunk");
            if (h_hdr[i].n > kInlinePeaks || (h_hdr[i].overflow & 2)) big = true;
            if (!h->have_min[sm] || h_hdr[i].seg_min < h->min_seg_min[sm]) { h->min_seg_min[sm] = h_hdr[i].seg_min; h->have_min[sm] = true; }
        }
        all.clear();
        if (!big) {
            // collect in window order (audio_matcher.rs:132-133)
            for (int i = s0; i < s1; ++i)
                for (int j = 0; j < h_hdr[i].n; ++j) all.push_back(h_hdr[i].first[j]);
        } else {
            // rare: a chunk with more peaks than a header holds (its full list was
            // overwritten by later haystacks) or a chunk whose minimum was below
            // what theta assumed: redo this haystack on its own, writing every score
            const long long out_count = (long long)(lens[k] - s + 1);
            ScanRequest full = scan;
            full.theta = -FLT_MAX;
            if ((rc = run_correlation(h, d_hays[k], (long long)lens[k], 0, (float*)c->scores.p, out_count, factor,
                                      &full, src_kind))) return rc;
            if ((rc = launch_pick(c, (const float*)c->scores.p, out_count, s0, s1 - s0, p->min_prominence,
                                  (long long)p->min_distance, &full))) return rc;
            AM_HIP(hipStreamSynchronize(c->stream));
            for (int i = s0; i < s1; ++i) {
                if (h_hdr[i].overflow & 1) return fail(AM_ERR_PEAK_OVERFLOW, "more than AM_MAX_PEAKS_PER_CHUNK peaks in one chunk");
                const int cnt = h_hdr[i].n;
                if (cnt <= 0) continue;
                const size_t old = all.size();
                all.resize(old + cnt);
                AM_HIP(copy_on_stream(c, all.data() + old, (am_peak*)c->peaks.p + (size_t)(i - s0) * AM_MAX_PEAKS_PER_CHUNK,
                                 sizeof(am_peak) * cnt, hipMemcpyDeviceToHost));
            }
        }
        rc = merge_peaks(all, p, out ? out + k * cap_per_hay : nullptr, cap_per_hay, &n_out[k]);
        if (g_progress_fn) g_progress_fn(g_progress_user, k, 1, (size_t)(s1 - s0));
        if (rc == AM_ERR_CAPACITY) worst = rc;
        else if (rc) return rc;
    }
    return worst;
}


// BASELINE config 4: several needles against one resident haystack.  The
// haystack's forward column pass (K1) runs once; every needle then gets its own
// K2 (row transforms + multiply with that needle's spectrum, written to a second
// work matrix), K3 (fused scan) and peak pick.  Needles must share one length so
// that they share the block layout.
static int match_multi(am_needle* const* needles, size_t nn, const void* d_hay, size_t len, int src_kind,
                       const am_match_params* p, am_peak* out, size_t cap_per_needle, size_t* n_out) {
    am_needle* h0 = needles[0];
    Ctx* c = h0->ctx;
    const size_t s = h0->n;
    for (size_t k = 0; k < nn; ++k) {
        n_out[k] = 0;
        if (!needles[k] || needles[k]->ctx != c) return fail(AM_ERR_INVALID_ARG, "needles must live on one device");
        if (needles[k]->n != s) return fail(AM_ERR_INVALID_ARG, "am_match_multi: needles must have equal length");
    }
    if (p->chunk == 0) return fail(AM_ERR_INVALID_ARG, "chunk must be > 0");
    if (p->scale != AM_SCALE_NONE && p->scale != AM_SCALE_LIB)
        return fail(AM_ERR_INVALID_ARG, "am_match supports AM_SCALE_NONE and AM_SCALE_LIB");
    if (len < s) return AM_OK;
    const int sm = p->scale == AM_SCALE_LIB ? 1 : 0;
    std::vector<Segment> segs;
    make_segments(len, s, p, segs);
    const int nsegs = (int)segs.size();
    if (nsegs == 0) return AM_OK;
    const long long out_count = (long long)(len - s + 1);
    int rc, logN = 0;
    if ((rc = pick_log_n(s, out_count, &logN))) return rc;
    const Plan* pl = nullptr;
    if ((rc = get_plan(c, logN, &pl))) return rc;
    const long long N = 1ll << logN;
    long long hop = N - (long long)s + 1;
    if (hop >= 8 * kTile) hop = (hop / kTile) * kTile;
    const long long nblocks = (out_count + hop - 1) / hop;
    const long long npairs = (nblocks + 1) / 2;
    std::vector<const float2*> hcs(nn);
    for (size_t k = 0; k < nn; ++k)
        if ((rc = needle_spectrum(needles[k], pl, &hcs[k]))) return rc;   // may use c->work: before it is filled
    if ((rc = c->work.ensure((size_t)npairs * (size_t)N * sizeof(float2)))) return rc;
    const bool half = g_opt_half && plan_is_r16(pl->dev) && !pl->dev.wide && g_k2_variant == 0;
    // needles are taken in groups that share the forward row transforms of K2
    const size_t group = (!half && plan_k2_has_group(pl->dev))
        ? (size_t)std::min<long long>(std::max<long long>(1, g_opt_needle_group), kMaxNeedleGroup) : 1;
    const size_t matrix = (size_t)npairs * (size_t)N;   // points of one needle's work matrix
    if ((rc = c->work2.ensure(std::min(group, nn) * matrix * sizeof(float2)))) return rc;
    if ((rc = c->scores.ensure((size_t)out_count * sizeof(float)))) return rc;
    const size_t hdr_bytes = sizeof(SegHeader) * nsegs * nn;
    if ((rc = c->hdr.ensure(hdr_bytes))) return rc;
    if ((rc = c->peaks.ensure(sizeof(am_peak) * (size_t)nsegs * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    if ((rc = upload_segments(c, segs))) return rc;
    SegHeader* h_hdr = static_cast<SegHeader*>(c->hdr.p);
    const bool fused = plan_is_r16(pl->dev) && (hop % kTile) == 0;
    if (fused) {
        if ((rc = c->stats32.ensure((size_t)((out_count + 31) / 32) * sizeof(float2)))) return rc;
        if ((rc = c->wflags.ensure((size_t)nblocks << (pl->dev.logN2 + pl->dev.wide - kColsLog)))) return rc;
    }
    Job job{};
    job.src = d_hay; job.src_len = (long long)len; job.lead = 0; job.src_kind = src_kind;
    job.dst = (float*)c->scores.p; job.out_count = out_count; job.hop = (int)hop; job.nblocks = (int)nblocks;
    job.first_pair = 0;
    { ProfScope ps(c, KN_K1); AM_HIP(launch_k1(c->stream, job, (int)npairs, (float2*)c->work.p, pl->dev, half)); }
    for (size_t k = 0; k < nn; ++k) {
        am_needle* h = needles[k];
        const size_t in_group = k % group;
        const float2* inv_rows = (const float2*)c->work2.p + in_group * matrix;   // this needle's inverse rows
        if (group > 1 && in_group == 0) {
            K2Group grp{};
            grp.n = (int)std::min(group, nn - k);
            for (int j = 0; j < grp.n; ++j) { grp.hc[j] = hcs[k + j]; grp.dst[j] = (float2*)c->work2.p + (size_t)j * matrix; }
            ProfScope ps(c, KN_K2);
            AM_HIP(launch_k2_group(c->stream, (int)npairs, (const float2*)c->work.p, grp, pl->dev));
        }
        ScanRequest scan{};
        scan.theta = (h->have_min[sm] && p->min_prominence > 0.f) ? h->min_seg_min[sm] + 0.5f * p->min_prominence : -FLT_MAX;
        scan.seg_c = (long long)p->chunk;
        scan.seg_d = (long long)(p->chunk + p->overlap) - (long long)s;
        scan.fused = fused;
        scan.sparse = SparseScores{nullptr, nullptr, 0.f, (int)hop, pl->dev.logN2 + pl->dev.wide, 1.0 / (double)hop};
        ScanCfg cfg{};
        if (fused) {
            cfg.stats32 = (float2*)c->stats32.p; cfg.wflags = (unsigned char*)c->wflags.p; cfg.theta = scan.theta;
            cfg.seg_c = scan.seg_c; cfg.seg_d = scan.seg_d; cfg.inv_c = 1.0 / (double)scan.seg_c;
            scan.sparse = SparseScores{cfg.wflags, cfg.stats32, cfg.theta, (int)hop, pl->dev.logN2 + pl->dev.wide, 1.0 / (double)hop};
        }
        const float factor = scale_factor(h, p->scale, 1);
        const float hscale = half ? kHalfGain * h->inv_autocorr : 1.0f;
        if (group == 1) { ProfScope ps(c, KN_K2); AM_HIP(launch_k2(c->stream, (int)npairs, (float2*)c->work.p, hcs[k], pl->dev, (float2*)c->work2.p, half, hscale)); }
        { ProfScope ps(c, KN_K3); AM_HIP(launch_k3(c->stream, job, (int)npairs, inv_rows, pl->dev,
                                                  half ? factor / hscale : factor, cfg, half)); }
        if ((rc = launch_pick(c, (const float*)c->scores.p, out_count, 0, nsegs, p->min_prominence,
                              (long long)p->min_distance, &scan, (int)(k * nsegs)))) return rc;
    }
    AM_HIP(hipStreamSynchronize(c->stream));
    int worst = AM_OK;
    std::vector<am_peak> all;
    for (size_t k = 0; k < nn; ++k) {
        am_needle* h = needles[k];
        const SegHeader* hd = h_hdr + k * nsegs;
        bool big = false;
        for (int i = 0; i < nsegs; ++i) {
            if (hd[i].overflow & 1) return fail(AM_ERR_PEAK_OVERFLOW, "more than AM_MAX_PEAKS_PER_CHUNK peaks in one chunk");
            if (hd[i].n > kInlinePeaks || (hd[i].overflow & 2)) big = true;
            if (!h->have_min[sm] || hd[i].seg_min < h->min_seg_min[sm]) { h->min_seg_min[sm] = hd[i].seg_min; h->have_min[sm] = true; }
        }
        am_peak* dst = out ? out + k * cap_per_needle : nullptr;
        if (big) {   // rare: redo this needle alone through the single-needle path (writes every score)
            const bool keep = h->have_min[sm];
            h->have_min[sm] = false;             // forces theta = -inf
            rc = match_many(h, &d_hay, &len, 1, p, dst, cap_per_needle, &n_out[k], src_kind);
            h->have_min[sm] = h->have_min[sm] || keep;
        } else {
            all.clear();
            for (int i = 0; i < nsegs; ++i)
                for (int j = 0; j < hd[i].n; ++j) all.push_back(hd[i].first[j]);
            rc = merge_peaks(all, p, dst, cap_per_needle, &n_out[k]);
        }
        if (rc == AM_ERR_CAPACITY) worst = rc;
        else if (rc) return rc;
    }
    return worst;
}

// find_peaks on one host score array (am_find_peaks)
static int find_peaks_host_array(Ctx* c, const float* d_scores, long long n, float min_prom, long long min_dist,
                                 std::vector<am_peak>& all) {
    int rc;
    Segment sg; sg.a = 0; sg.b = n;
    if ((rc = c->hdr.ensure(sizeof(SegHeader)))) return rc;
    if ((rc = c->peaks.ensure(sizeof(am_peak) * AM_MAX_PEAKS_PER_CHUNK))) return rc;
    if ((rc = upload_segments(c, std::vector<Segment>(1, sg)))) return rc;
    if ((rc = launch_pick(c, d_scores, n, 0, 1, min_prom, min_dist, nullptr))) return rc;
    AM_HIP(hipStreamSynchronize(c->stream));
    const SegHeader hd = *static_cast<const SegHeader*>(c->hdr.p);
    if (hd.overflow) return fail(AM_ERR_PEAK_OVERFLOW, "more than AM_MAX_PEAKS_PER_CHUNK peaks in one chunk");
    all.resize(hd.n);
    if (hd.n > 0) AM_HIP(copy_on_stream(c, all.data(), c->peaks.p, sizeof(am_peak) * hd.n, hipMemcpyDeviceToHost));
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
        for (auto& kv : h->spectra) (void)hipFree(kv.second);
        if (h->d_needle) (void)hipFree(h->d_needle);
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
    if ((rc = run_correlation(h, d_in, (long long)w, lead, d_out, (long long)len, scale_factor(h, scale, w)))) return rc;
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
    return match_multi(const_cast<am_needle* const*>(needles), n_needles, d_haystack, len, 0, p, out, cap_per_needle, n_out);
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

int am_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto& kv : g_ctx) {
        Ctx* c = kv.second;
        std::lock_guard<std::recursive_mutex> lk2(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        for (DevBuf* b : {&c->work, &c->work2, &c->scores, &c->stats, &c->stats32, &c->wflags, &c->segs,
                          &c->scores_b, &c->stats_b, &c->stats32_b, &c->wflags_b, &c->peaks_b,
                          &c->peaks, &c->io_in, &c->io_out, &c->sum})
            b->release();
        if (c->pinned.p) { (void)hipHostFree(c->pinned.p); c->pinned.p = nullptr; c->pinned.cap = 0; }
        if (c->hdr.p) { (void)hipHostFree(c->hdr.p); c->hdr.p = nullptr; c->hdr.cap = 0; }
        c->segs_resident.clear();
        for (auto& pk : c->plans) if (pk.second.tables) (void)hipFree(pk.second.tables);
        c->plans.clear();
        for (auto& r : c->pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        c->pending.clear();
        for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
        c->pool.clear();
    }
    return AM_OK;
}

int am_set_progress_callback(am_progress_fn fn, void* user) {
    g_progress_fn = fn;
    g_progress_user = user;
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
    if (!strcmp(key, "half_pipeline")) { g_opt_half = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "batch_overlap")) { g_opt_batch_overlap = value ? 1 : 0; return AM_OK; }
    if (!strcmp(key, "vmm_work")) { g_opt_vmm = value < 0 ? 0 : value; return AM_OK; }
    if (!strcmp(key, "needle_group")) {
        if (value < 1 || value > kMaxNeedleGroup) return fail(AM_ERR_INVALID_ARG, "needle_group out of range");
        g_opt_needle_group = value; return AM_OK;
    }
    if (!strcmp(key, "profile_mask")) { g_opt_profile_mask = value; return AM_OK; }
    if (!strcmp(key, "lanes")) {
        if (value != 1 && value != 2) return fail(AM_ERR_INVALID_ARG, "lanes must be 1 or 2");
        g_opt_lanes = value; return AM_OK;
    }
    if (!strcmp(key, "k2_variant")) {
        if (value < 0 || value > 1) return fail(AM_ERR_INVALID_ARG, "k2_variant must be 0 or 1");
        g_k2_variant = (int)value; return AM_OK;
    }
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
    if (!strcmp(key, "k2_variant")) { *value = g_k2_variant; return AM_OK; }
    if (!strcmp(key, "half_pipeline")) { *value = g_opt_half; return AM_OK; }
    if (!strcmp(key, "needle_group")) { *value = g_opt_needle_group; return AM_OK; }
    if (!strcmp(key, "batch_overlap")) { *value = g_opt_batch_overlap; return AM_OK; }
    if (!strcmp(key, "lanes")) { *value = g_opt_lanes; return AM_OK; }
    if (!strcmp(key, "profile_mask")) { *value = g_opt_profile_mask; return AM_OK; }
    return fail(AM_ERR_INVALID_ARG, "unknown option");
}

}  // extern "C"
