/*
 * oracle/oracle.h -- CPU restatement of the reference's matcher hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * include, link, load or execute anything under oracle/.  The shipped library
 * (audio-matcher_amd/csrc -> libaudiomatch_amd.so) never calls into it and has
 * no CPU fallback.
 *
 * What is restated (all citations relative to /root/reference):
 *   - PCM down-mix              src/matcher/mp3_reader.rs:12, 28-37
 *   - correlation + crop        src/matcher/audio_matcher.rs:232-272, 414-464
 *   - production scaling        src/matcher/audio_matcher.rs:297-310, 321-329
 *   - chunk driver / merge      src/matcher/audio_matcher.rs:88-160, 221-230
 *   - offset restore            src/lib.rs:8-10
 *   - peak -> duration          src/matcher/mod.rs:127-129
 *
 * Pinning status:
 *   - The reference is a Rust crate; no Rust toolchain exists in the build
 *     container and its git/crates.io dependencies are unreachable, so there
 *     is no oracle/_ref build ("unbuildable here").
 *   - Correlation (unscaled, Valid): PINNED by the reference's own
 *     known-answer test audio_matcher.rs:490-517 (fixture K1) and the bench
 *     shape benches/my_benchmark.rs:31-32 (fixture K5).
 *   - Peak prominence values + output order: PINNED by
 *     audio_matcher.rs:167-185 (fixture K2).  Overshadow rule: PINNED by
 *     audio_matcher.rs:187-218 (fixture K3).
 *   - PARITY UNPINNED (third-party sources absent, no reference test covers
 *     them): find_peaks 0.1 min_distance semantics / plateau handling / filter
 *     order; common::chunked tail windows; common::filter_surrounding
 *     neighbour choice; fftconvolve 0.1 behaviour for within shorter than the
 *     needle.  The choices made are documented at each function.
 */
#ifndef AUDIOMATCH_ORACLE_H
#define AUDIOMATCH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mode (audio_matcher.rs:55-59) */
enum { ORC_MODE_FULL = 0, ORC_MODE_SAME = 1, ORC_MODE_VALID = 2 };

/* scale argument:
 *   0 = unscaled
 *   1 = LibConvolve (production, matcher/mod.rs:34): corr * 1/sum(needle^2)
 *       (audio_matcher.rs:306-308, 321-329)
 *   2 = MyConvolve: corr * 1/sum(needle^2) / within.len()
 *       (audio_matcher.rs:442-448) */
enum { ORC_SCALE_NONE = 0, ORC_SCALE_LIB = 1, ORC_SCALE_MY = 2 };

/* FFT length policy of orc_correlate */
enum {
    ORC_FFT_REFERENCE = 0, /* pad_len = w+s-1 exactly as audio_matcher.rs:421 (Bluestein if needed) */
    ORC_FFT_POW2 = 1,      /* pad_len = next power of two (same result, faster checker) */
    ORC_FFT_DIRECT = 2,    /* O(w*s) direct summation in f64 (tiny shapes, exact) */
    ORC_FFT_POW2_CACHED = 3 /* orc_calc_chunks only: ORC_FFT_POW2 with the needle spectrum computed
                              once per call and shared by all full-length windows */
};
/* arithmetic of the transforms */
enum { ORC_PREC_F64 = 0, ORC_PREC_F32 = 1 };

typedef struct {
    uint64_t start;      /* find_peaks::Peak::position.start (absolute after offset_range) */
    uint64_t end;        /* position.end (exclusive; plateau) */
    float height;
    float prominence;
} orc_peak;

/* The rules no source or test available offline pins (find_peaks 0.1 and common are absent from the reference
 * tree).  A zeroed struct (or NULL) = the documented defaults; every alternative is restated in
 * oracle.c and implemented in the library as the option of the same name (am_set_option).
 *   peak_filter_order  0: prominence filter, then distance filter       1: distance first, then prominence (scipy's order)
 *   distance_rule      bit 0 -- a peak is dropped at a distance  0: < min_distance   1: <= min_distance  from a kept, higher one
 *                      bit 1 -- measured between                0: plateau middles  1: plateau starts
 *   tail_window        0: chunked() emits the shorter windows at the end of the haystack   1: full-length windows only
 *   surrounding_from   filter_surrounding's neighbours: 0: both from the sorted, unfiltered sequence
 *                                                       1: the one before = the last element kept (sequential filter) */
typedef struct {
    int peak_filter_order;
    int distance_rule;
    int tail_window;
    int surrounding_from;
} orc_policy;

/* mp3_reader.rs:28-37: mono = (l as f32 + r as f32) * 0.5 * PCM_FACTOR,
 * PCM_FACTOR = 1.0 / 65535 as f32 (mp3_reader.rs:12). */
void orc_pcm_s16_stereo_to_mono(const int16_t* interleaved_lr, size_t frames, float* out);

/* output length of a mode for (w, s): audio_matcher.rs:450-456 */
size_t orc_mode_len(size_t w, size_t s, int mode);
/* start offset of the crop inside the full (w+s-1) array: audio_matcher.rs:460-464 */
size_t orc_mode_start(size_t w, size_t s, int mode);

/* 1 / sum(needle^2): audio_matcher.rs:321-329 / 404-413 (element 0 of the
 * needle's Valid correlation with itself, then reciprocal), in f64 -> f32 */
float orc_inv_autocorr(const float* needle, size_t s);

/* CorrelateAlgo::correlate_with_sample.  Returns number of outputs written,
 * or (size_t)-1 on allocation failure / cap too small. */
size_t orc_correlate(const float* within, size_t w, const float* needle, size_t s,
                     int mode, int scale, int fft_policy, int precision,
                     float* out, size_t cap);

/* find_peaks::PeakFinder::new(y).with_min_prominence(p).with_min_distance(d).find_peaks()
 * (audio_matcher.rs:221-230).  Positions are local to y.  Output order: by
 * height descending (pinned by K2's order), ties by position ascending.
 * Returns the number of peaks (may exceed cap; only cap are written). */
size_t orc_find_peaks(const float* y, size_t n, float min_prominence, size_t min_distance,
                      orc_peak* out, size_t cap);

size_t orc_find_peaks_policy(const float* y, size_t n, float min_prominence, size_t min_distance,
                             const orc_policy* pol, orc_peak* out, size_t cap);

/* is_overshadowed (audio_matcher.rs:143-160); other == NULL is Option::None */
int orc_is_overshadowed(const orc_peak* element, const orc_peak* other, uint32_t sr,
                        double max_distance_s);

/* calc_chunks (audio_matcher.rs:88-141).  chunk / overlap are already in
 * samples (:99-100 do the rounding from durations: use orc_round_samples).
 * min_distance is in samples (audio_matcher.rs:228), overshadow distance in
 * seconds (:137-138).  threads > 1 fans chunks out over pthreads, mirroring
 * par_bridge (:114).  Returns number of peaks (sorted by start, filtered). */
size_t orc_calc_chunks(uint32_t sr, const float* haystack, size_t h,
                       const float* needle, size_t s,
                       size_t chunk, size_t overlap,
                       float min_prominence, size_t min_distance, double overshadow_distance_s,
                       int scale, int fft_policy, int precision, int threads,
                       orc_peak* out, size_t cap);

size_t orc_calc_chunks_policy(uint32_t sr, const float* haystack, size_t h,
                              const float* needle, size_t s,
                              size_t chunk, size_t overlap,
                              float min_prominence, size_t min_distance, double overshadow_distance_s,
                              int scale, int fft_policy, int precision, int threads, const orc_policy* pol,
                              orc_peak* out, size_t cap);

/* (secs * sr).round() as usize : audio_matcher.rs:99-100 */
size_t orc_round_samples(double seconds, uint32_t sr);

/* Duration::from_secs_f64(start / sr) in integer nanoseconds (matcher/mod.rs:127-129) */
uint64_t orc_start_nanos(uint64_t start, uint32_t sr);

/* counter-based synthetic signal: uniform in [-amp, amp),
 * exactly representable in f32 */
void orc_synth_uniform(uint32_t seed, uint32_t stream, uint64_t first, size_t n, float amp, float* out);

#ifdef __cplusplus
}
#endif
#endif
