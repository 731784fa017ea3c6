// pushbench -- what feeding the library from host memory costs (C ABI only; the numbers of
// bench.py's "host_feed" side leg and profiles/r04/host_feed.json).
//
// One synthetic 1 h haystack (44.1 kHz f32 mono, 6 planted copies of a 10 s needle) is
//   * matched from a pageable and from a pinned buffer (am_match), and
//   * pushed through am_match_stream_begin / push / finish in pieces of 8 M samples, 64 K samples and the
//     decoder's 1152 samples (minimp3's frame, mp3_reader.rs:28-37), from pageable and from pinned memory.
// Reported per leg: wall seconds, samples/s, and the CPU time of the pushing thread per GB pushed
// (CLOCK_THREAD_CPUTIME_ID) -- what a decoder thread loses to the hand-over.  Offsets are checked every time.
#include <time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/audiomatch.h"

static double now(clockid_t c) { timespec t; clock_gettime(c, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: %d %s\n", #x, rc_, am_last_error_string()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int hours_den = argc > 1 ? atoi(argv[1]) : 1;      // 1 = a whole hour, 4 = a quarter ...
    const uint32_t sr = 44100;
    const size_t s = 10 * sr, h = (size_t)3600 * sr / (size_t)(hours_den > 0 ? hours_den : 1);
    int ndev = 0;
    CK(am_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    // synthetic signal on the device, fetched to the host once
    void *d_needle = nullptr, *d_hay = nullptr;
    CK(am_device_malloc(0, s * 4, &d_needle));
    CK(am_device_malloc(0, h * 4, &d_hay));
    CK(am_synth_uniform_device(0, (float*)d_needle, 1, 0, 0, s, 0.25f));
    CK(am_synth_uniform_device(0, (float*)d_hay, 1, 1, 0, h, 0.25f));
    std::vector<uint64_t> plants;
    for (int m = 0; m < 6; ++m) {
        const uint64_t t = (uint64_t)600 * sr * m + 30 * sr + 1234;
        if (t + s <= h) { plants.push_back(t); CK(am_axpy_device(0, (float*)d_hay + t, (const float*)d_needle, s, 1.0f)); }
    }
    am_needle* needle = nullptr;
    CK(am_needle_create_device(0, (const float*)d_needle, s, &needle));
    std::vector<float> pageable(h);
    CK(am_memcpy_d2h(0, pageable.data(), d_hay, h * 4));
    float* pinned = nullptr;
    CK(am_host_alloc(h * 4, (void**)&pinned));
    memcpy(pinned, pageable.data(), h * 4);
    am_match_params p{};
    p.sr = sr; p.chunk = 60 * sr; p.overlap = s; p.min_prominence = 0.13f; p.min_distance = (uint64_t)480 * sr;
    p.overshadow_distance_s = 480.0; p.scale = AM_SCALE_LIB;
    am_peak out[64]; size_t n = 0;
    auto check = [&]() {
        if (n != plants.size()) return false;
        for (size_t i = 0; i < n; ++i) if (out[i].start != plants[i]) return false;
        return true;
    };
    printf("{\"haystack_samples\": %zu, \"legs\": [", h);
    bool first = true;
    auto report = [&](const char* name, const char* mem, size_t piece, double wall, double cpu, bool ok) {
        printf("%s\n  {\"leg\": \"%s\", \"memory\": \"%s\", \"piece_samples\": %zu, \"wall_s\": %.6f, \"samples_per_s\": %.4g, "
               "\"push_thread_cpu_s_per_GB\": %.4f, \"offsets_ok\": %s}", first ? "" : ",", name, mem, piece, wall, h / wall,
               cpu / (h * 4 / 1e9), ok ? "true" : "false");
        first = false;
    };
    for (int mem = 0; mem < 2; ++mem) {
        const float* src = mem ? pinned : pageable.data();
        const char* mname = mem ? "pinned (am_host_alloc)" : "pageable";
        for (int rep = 0; rep < 2; ++rep) {            // (the second repetition is reported)
            const double w0 = now(CLOCK_MONOTONIC), c0 = now(CLOCK_THREAD_CPUTIME_ID);
            CK(am_match(needle, src, h, &p, out, 64, &n));
            const double w1 = now(CLOCK_MONOTONIC), c1 = now(CLOCK_THREAD_CPUTIME_ID);
            if (rep) report("am_match", mname, h, w1 - w0, c1 - c0, check());
        }
        am_stream* st = nullptr;
        CK(am_match_stream_begin(needle, AM_FMT_F32_MONO, h, &p, &st));
        const size_t pieces[3] = {(size_t)8 << 20, (size_t)64 << 10, 1152};
        for (size_t piece : pieces) {
            for (int rep = 0; rep < 2; ++rep) {
                const double w0 = now(CLOCK_MONOTONIC), c0 = now(CLOCK_THREAD_CPUTIME_ID);
                for (size_t off = 0; off < h; off += piece) CK(am_match_stream_push(st, src + off, off + piece <= h ? piece : h - off));
                const double c1 = now(CLOCK_THREAD_CPUTIME_ID);
                CK(am_match_stream_finish(st, out, 64, &n));
                const double w1 = now(CLOCK_MONOTONIC);
                if (rep) report("am_match_stream_push", mname, piece, w1 - w0, c1 - c0, check());
            }
        }
        am_match_stream_destroy(st);
    }
    printf("\n]}\n");
    am_needle_destroy(needle);
    am_host_free(pinned);
    am_device_free(0, d_needle); am_device_free(0, d_hay);
    return 0;
}
