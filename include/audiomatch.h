/*
 * audiomatch.h -- C ABI of libaudiomatch_amd.so, the MI355X (gfx950) native
 * implementation of NilsJochem/audio-matcher's src/matcher hot path:
 * sliding-window FFT cross-correlation + prominence peak pick.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, never
 * throws or aborts across the boundary, and returns an int status
 * (AM_OK == 0).  am_last_error_string() gives a thread-local description of
 * the last failure on the calling thread.
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to the reference crate root).
 *
 * There is NO CPU fallback: with no usable HIP device every compute entry
 * point returns AM_ERR_NO_DEVICE / AM_ERR_HIP.
 */
#ifndef AUDIOMATCH_H
#define AUDIOMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AM_ABI_VERSION 1

/* status codes */
enum {
    AM_OK = 0,
    AM_ERR_INVALID_ARG = 1,   /* null pointer, zero length, unsupported size ... */
    AM_ERR_CAPACITY = 2,      /* caller buffer too small; required length was written */
    AM_ERR_HIP = 3,           /* a HIP runtime call or kernel launch failed */
    AM_ERR_NO_DEVICE = 4,     /* no gfx950-capable device / bad ordinal */
    AM_ERR_PEAK_OVERFLOW = 5, /* more than AM_MAX_PEAKS_PER_CHUNK peaks in one chunk */
    AM_ERR_OOM = 6            /* host or device allocation failed */
};

/* audio_matcher.rs:55-59  enum Mode { Full, Same, Valid } */
enum { AM_MODE_FULL = 0, AM_MODE_SAME = 1, AM_MODE_VALID = 2 };

/* `scale: bool` of CorrelateAlgo::correlate_with_sample (audio_matcher.rs:67-72),
 * split by implementation because the two reference algos disagree
 * (SURVEY.md F4):
 *   AM_SCALE_LIB = LibConvolve, the production algo (matcher/mod.rs:34):
 *                  corr / sum(needle^2)            audio_matcher.rs:306-308
 *   AM_SCALE_MY  = MyConvolve: corr / sum(needle^2) / within.len()
 *                                                  audio_matcher.rs:442-448 */
enum { AM_SCALE_NONE = 0, AM_SCALE_LIB = 1, AM_SCALE_MY = 2 };

/* Upper bound on peaks that pass the prominence filter inside ONE chunk. */
#define AM_MAX_PEAKS_PER_CHUNK 1024

/* Opaque handle = the reference's `LibConvolve { sample_data, .. }` /
 * `MyConvolve` object (audio_matcher.rs:282-295, 379-403): owns a device copy
 * of the needle, its cached 1/sum(needle^2) and its cached spectra. */
typedef struct am_needle am_needle;

/* find_peaks::Peak<f32> as consumed downstream (only position and prominence
 * are read: matcher/mod.rs:110-129, archive/data.rs:87-107). */
typedef struct am_peak {
    uint64_t start;    /* position.start, absolute sample offset in the haystack */
    uint64_t end;      /* position.end (exclusive) */
    float height;      /* score at the peak */
    float prominence;  /* Option<f32>, always Some on this path */
} am_peak;

/* Parameters of calc_chunks (audio_matcher.rs:88-97) after Config::from_args
 * (audio_matcher.rs:38-52) and the duration->sample rounding of :99-100. */
typedef struct am_match_params {
    uint32_t sr;                  /* sample rate (u16 in the reference) */
    uint64_t chunk;               /* chunk_size in samples   (:100) */
    uint64_t overlap;             /* overlap_length in samples (:99) */
    float min_prominence;         /* PeakConfig.prominence = args.prominence/100 (:44) */
    uint64_t min_distance;        /* distance.as_secs() * sr, samples (:228) */
    double overshadow_distance_s; /* PeakConfig.distance in seconds (:137-138) */
    int scale;                    /* AM_SCALE_*; production passes true -> AM_SCALE_LIB (mod.rs:85) */
} am_match_params;

/* ---- library / device ------------------------------------------------- */
int am_abi_version(void);
const char* am_last_error_string(void);
int am_device_count(int* n);
/* Releases every scratch buffer, transform plan and timing event the library
 * holds on every device (needle handles stay valid and rebuild what they need
 * on their next use).  Optional: everything is also released at process exit. */
int am_shutdown(void);

/* Device pointers (every `d_` argument, and everything the `_device` entry
 * points take): the library works on a stream of its own, which is NOT ordered
 * with the caller's streams or with the null stream.  Data behind a device
 * pointer must be complete before the call (synchronise whatever produced it;
 * note that a device-to-device hipMemcpy returns before the copy has run), and
 * results written to a device pointer are complete when the call returns. */

/* ---- needle handle ----------------------------------------------------- */
/* LibConvolve::new(sample_data) audio_matcher.rs:289 / MyConvolve::new :396.
 * `needle` is host memory, copied; the handle lives on `device`. */
int am_needle_create(int device, const float* needle, size_t n, am_needle** out);
/* same, needle already resident on `device` */
int am_needle_create_device(int device, const float* d_needle, size_t n, am_needle** out);
void am_needle_destroy(am_needle* h);
int am_needle_len(const am_needle* h, size_t* n);
/* CorrelateAlgo::inverse_sample_auto_correlation (audio_matcher.rs:66, 321-329) */
int am_needle_inv_autocorr(const am_needle* h, float* out);

/* ---- level 1: one correlation (the trait method) ------------------------ */
/* output length of a mode: audio_matcher.rs:450-456 */
int am_correlate_len(size_t w, size_t s, int mode, size_t* out_len);
/* CorrelateAlgo::correlate_with_sample(&self, within, mode, scale)
 * (audio_matcher.rs:67-72, 331-343, 471-478).  Host buffers in and out.
 * On AM_ERR_CAPACITY *out_len holds the required length. */
int am_correlate(const am_needle* h, const float* within, size_t w, int mode, int scale,
                 float* out, size_t cap, size_t* out_len);
/* same with device-resident input and output */
int am_correlate_device(const am_needle* h, const float* d_within, size_t w, int mode, int scale,
                        float* d_out, size_t cap, size_t* out_len);

/* ---- level 2: the chunked matcher --------------------------------------- */
/* calc_chunks(sr, m_samples, &algo, scale, config) (audio_matcher.rs:88-141):
 * windowing, per-chunk Valid correlation, per-chunk find_peaks
 * (:221-230), offset restore (:126), sort by start (:135) and the overshadow
 * filter (:136-139, 143-160).  Returns peaks sorted by start.
 * On AM_ERR_CAPACITY *n_out holds the number of peaks found. */
int am_match(const am_needle* h, const float* haystack, size_t len,
             const am_match_params* p, am_peak* out, size_t cap, size_t* n_out);
/* haystack already resident in HBM */
int am_match_device(const am_needle* h, const float* d_haystack, size_t len,
                    const am_match_params* p, am_peak* out, size_t cap, size_t* n_out);
/* The per-file loop of matcher::run (matcher/mod.rs:42-87) over resident
 * haystacks: out holds cap_per_hay slots per haystack, n_out[k] the count
 * for haystack k (if n_out[k] > cap_per_hay the call returns AM_ERR_CAPACITY
 * after filling what fits). */
int am_match_batch_device(const am_needle* h, const float* const* d_haystacks, const size_t* lens,
                          size_t n_hay, const am_match_params* p,
                          am_peak* out, size_t cap_per_hay, size_t* n_out);

/* Several needles (equal length, same device) against one resident haystack
 * (BASELINE config 4): the haystack's forward column pass is computed once and
 * its forward row transforms once per group of needles (option "needle_group");
 * out holds cap_per_needle slots per needle, n_out[k] the count for needle k.
 * The reference has no such entry point (one snippet per run,
 * matcher/mod.rs:29-34); offsets equal those of n_needles separate
 * am_match_device calls, heights and prominences agree to f32 rounding. */
int am_match_multi_device(const am_needle* const* needles, size_t n_needles, const float* d_haystack, size_t len,
                          const am_match_params* p, am_peak* out, size_t cap_per_needle, size_t* n_out);

/* The same matcher on interleaved 16-bit stereo PCM, the sample format the
 * reference decodes to (mp3_reader.rs:26 asserts two channels): the down-mix
 * mono = (l as f32 + r as f32) * 0.5 * (1/65535) (mp3_reader.rs:12, 28-37) is
 * fused into the first kernel's loads, bit-exact, so PCM is read once. */
int am_needle_create_pcm16(int device, const int16_t* interleaved, size_t frames, am_needle** out);
int am_match_pcm16(const am_needle* h, const int16_t* interleaved, size_t frames,
                   const am_match_params* p, am_peak* out, size_t cap, size_t* n_out);
int am_match_pcm16_device(const am_needle* h, const int16_t* d_interleaved, size_t frames,
                          const am_match_params* p, am_peak* out, size_t cap, size_t* n_out);
int am_match_pcm16_batch_device(const am_needle* h, const int16_t* const* d_interleaved, const size_t* frames,
                                size_t n_hay, const am_match_params* p,
                                am_peak* out, size_t cap_per_hay, size_t* n_out);

/* find_peaks(y_data, sr, PeakConfig) (audio_matcher.rs:221-230) =
 * PeakFinder::new(y).with_min_prominence(p).with_min_distance(d).find_peaks()
 * on one host score array; peaks come back by descending height. */
int am_find_peaks(int device, const float* scores, size_t n, float min_prominence,
                  uint64_t min_distance, am_peak* out, size_t cap, size_t* n_out);

/* ---- ingest: PCM -> f32 mono -------------------------------------------- */
/* mp3_reader.rs:12, 28-37: mono = (l as f32 + r as f32) * 0.5 * (1/65535) */
int am_pcm_s16_stereo_to_mono(int device, const int16_t* interleaved, size_t frames, float* out);
int am_pcm_s16_stereo_to_mono_device(int device, const int16_t* d_interleaved, size_t frames,
                                     float* d_out);

/* ---- device memory plumbing (for hosts without their own HIP allocator) -- */
int am_device_malloc(int device, size_t bytes, void** out);
int am_device_free(int device, void* p);
int am_memcpy_h2d(int device, void* d_dst, const void* src, size_t bytes);
int am_memcpy_d2h(int device, void* dst, const void* d_src, size_t bytes);
int am_device_synchronize(int device);

/* ---- synthetic signals for tests / benches -------- */
/* d_out[k] = uniform(seed, stream, first + k) * amp, 24-bit exact in [-amp, amp) */
int am_synth_uniform_device(int device, float* d_out, uint32_t seed, uint32_t stream,
                            uint64_t first, size_t n, float amp);
/* d_dst[i] += gain * d_src[i]  (plants a needle into a haystack) */
int am_axpy_device(int device, float* d_dst, const float* d_src, size_t n, float gain);

/* ---- progress hook ---------------------------------------------------------- */
/* The two-stage progress callbacks of calc_chunks (audio_matcher.rs:102-117, 129:
 * f1 when a chunk is picked up, f2 when it is done).  All chunks of a haystack
 * run in one set of launches here, so the hook fires per haystack: stage 0 with
 * its chunk count when it is queued, stage 1 when its peaks are back on the host.
 * Process-wide; pass NULL to clear. */
typedef void (*am_progress_fn)(void* user, size_t haystack_index, int stage, size_t n_chunks);
int am_set_progress_callback(am_progress_fn fn, void* user);

/* ---- measurement hooks ---------------------------------------------------- */
/* When enabled every kernel launch of the pipeline on `device` is bracketed by
 * HIP events on the stream it is launched on.  am_profile_query returns the
 * summed elapsed time and launch count of kernels whose name matches `kernel`
 * exactly ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats",
 * "peaks", or "*" for all). */
int am_profile_enable(int device, int on);
int am_profile_reset(int device);
int am_profile_query(int device, const char* kernel, double* total_ms, uint64_t* launches);

/* options:
 *   "log_n" (0 = auto), "pairs_per_group", "k2_variant" (0/1), "lanes" (1/2),
 *       "profile_mask": tuning / measurement knobs
 *   "batch_overlap" (0/1, default 1): in am_match_batch_device pick the peaks of haystack k
 *       on a second stream beside the transforms of haystack k+1
 *   "vmm_work" (0 = off, else MB per physical chunk): measurement only; backs the work
 *       matrix with hipMemCreate chunks of that size (fragment size)
 *   "needle_group" (1..8, default 8): how many needles of am_match_multi_device share
 *       one forward row transform of the haystack (1 = one row pass per needle)
 *   "half_pipeline" (0/1): store the transform's work matrix in half precision
 *       (BASELINE config 5).  Butterflies stay f32; scores then carry an absolute
 *       error of about 1e-5 of the chunk's score range, hit offsets are unaffected. */
int am_set_option(const char* key, long long value);
int am_get_option(const char* key, long long* value);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOMATCH_H */
