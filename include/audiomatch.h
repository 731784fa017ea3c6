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

#define AM_ABI_VERSION 3

/* status codes */
enum {
    AM_OK = 0,
    AM_ERR_INVALID_ARG = 1,   /* null pointer, zero length, unsupported size ... */
    AM_ERR_CAPACITY = 2,      /* caller buffer too small; required length was written */
    AM_ERR_HIP = 3,           /* a HIP runtime call or kernel launch failed */
    AM_ERR_NO_DEVICE = 4,     /* no gfx950-capable device / bad ordinal */
    AM_ERR_PEAK_OVERFLOW = 5, /* a chunk of 2^32 scores or more with more than AM_MAX_PEAKS_PER_CHUNK peaks */
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

/* Sample format of a haystack buffer.  The reference decodes every file to interleaved 16-bit
 * stereo (mp3_reader.rs:26 asserts two channels) and hands calc_chunks the down-mixed f32 mono
 * stream (mp3_reader.rs:28-37); both ends of that step are accepted.  One element (an f32 sample,
 * or one stereo frame of two i16) is 4 bytes in either format. */
enum { AM_FMT_F32_MONO = 0, AM_FMT_S16_STEREO = 1 };

/* Not a limit on results: any number of peaks may pass the prominence filter in a chunk and, as
 * find_peaks does, the library returns every one the distance filter keeps (the caller's `cap`
 * is the only bound).  Up to this many per chunk are ordered and filtered on chip; a chunk with
 * more (a min_distance shorter than the chunk and a tiny prominence bound) takes a slower path
 * through a list in device memory. */
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
 * p->scale takes any AM_SCALE_* value, as the reference's generic calc_chunks takes any
 * CorrelateAlgo: with AM_SCALE_MY every window is scaled by its own
 * 1 / (sum(needle^2) * within.len()) (audio_matcher.rs:442-448).
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

/* The per-file loop of matcher::run (matcher/mod.rs:42-87) around SEVERAL snippets (BASELINE
 * config 4: "32 needles vs 1000 x 1 h haystacks, haystack FFT reused"): n_needles equal-length
 * needles on one device against n_hay resident haystacks of `sample_format` (AM_FMT_*; lens in
 * samples / frames).  Per haystack the forward column pass runs once and the forward row
 * transforms once per group of needles; the peak pick of one (haystack, needle) pair runs beside the
 * transforms of the next.  out holds cap_per_pair slots per pair, pair (haystack k, needle j) at
 * index k * n_needles + j; n_out likewise.  Results equal n_hay * n_needles separate am_match_device /
 * am_match_pcm16_device calls (offsets identical, heights and prominences to f32 rounding); a
 * haystack with non-finite samples loses exactly the windows that hold them, as there. */
int am_match_multi_batch_device(const am_needle* const* needles, size_t n_needles, const void* const* d_haystacks,
                                const size_t* lens, size_t n_hay, int sample_format, const am_match_params* p,
                                am_peak* out, size_t cap_per_pair, size_t* n_out);

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

/* ---- streaming ingest ---------------------------------------------------------- */
/* calc_chunks consumes a lazy ExactSizeIterator<Item = f32> (audio_matcher.rs:88-97): the decoder
 * yields frames (mp3_reader.rs:13-41) and the windows are cut as they arrive (:104).  The same
 * here: begin a stream (expected_len = the iterator's size hint, mp3_duration x sample rate,
 * matcher/mod.rs:77-83; 0 = unknown), push blocks of `sample_format` samples from host memory as
 * the decoder produces them, finish.  Every push is copied on a copy stream and the transforms of
 * every block pair whose samples have arrived completely are launched at once, so copying (or
 * decoding) and matching overlap; finish runs what is left, picks the peaks and returns exactly
 * what am_match / am_match_pcm16 return for the concatenated samples, bit for bit.  After finish
 * the stream is empty again and can take the next file.  One producer per stream; streams on one
 * device share its queue.
 * A push never waits for the device when the piece is small (below 1 MB: the decoder's 1152-frame pieces,
 * mp3_reader.rs:28-37): it is a host memcpy into a two-slot pinned staging ring, and a full slot (4 MB) leaves as
 * one asynchronous copy while the other fills.  A larger piece is copied straight from the caller's buffer -- at link
 * speed when that buffer is pinned (am_host_alloc / am_host_register) -- and push returns when the copy has left it.
 * `samples` may be reused as soon as push returns, either way. */
typedef struct am_stream am_stream;
int am_match_stream_begin(const am_needle* h, int sample_format, size_t expected_len, const am_match_params* p, am_stream** out);
int am_match_stream_push(am_stream* st, const void* samples, size_t n);
int am_match_stream_finish(am_stream* st, am_peak* out, size_t cap, size_t* n_out);
void am_match_stream_destroy(am_stream* st);

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

/* Pinned host memory for the buffers handed to am_match, am_match_stream_push and am_pool_match_* (the reference
 * collects the decoder's output in ordinary Vecs, mp3_reader.rs:13-41, matcher/mod.rs:32; nothing to pin there).
 * The copy engines read pinned memory directly: no bounce buffer inside the runtime and no page faults, so the
 * copier threads of a pool feed their devices side by side ("feeding the GPUs").  Either
 * allocate the decoder's output buffer here, or register an existing allocation for the time it is in use.
 * Portable across devices. */
int am_host_alloc(size_t bytes, void** out);
int am_host_free(void* p);
int am_host_register(void* p, size_t bytes);
int am_host_unregister(void* p);

/* ---- synthetic signals for tests / benches -------- */
/* d_out[k] = uniform(seed, stream, first + k) * amp, 24-bit exact in [-amp, amp) */
int am_synth_uniform_device(int device, float* d_out, uint32_t seed, uint32_t stream,
                            uint64_t first, size_t n, float amp);
/* d_dst[i] += gain * d_src[i]  (plants a needle into a haystack) */
int am_axpy_device(int device, float* d_dst, const float* d_src, size_t n, float gain);

/* the same signal as interleaved i16 stereo frames: left = stream,
 * right = stream + 5000, each value rint(uniform * amp * 32767), saturated */
int am_synth_pcm16_stereo_device(int device, int16_t* d_out, uint32_t seed, uint32_t stream,
                                 uint64_t first, size_t frames, float amp);
/* d_dst[i] = saturate(d_dst[i] + d_src[i]) over 2 * frames interleaved i16 values (plants a needle) */
int am_add_pcm16_device(int device, int16_t* d_dst, const int16_t* d_src, size_t frames);

/* ---- the haystack batch over every GPU of a node ----------------------------- */
/* matcher::run's per-file loop (matcher/mod.rs:42-87) sharded over devices: haystacks are
 * independent given the needle (audio_matcher.rs:114-131), so haystack k goes to pool slot
 * k mod n_dev -- no exchange step, no collective; results are gathered on the host.
 * am_shard_plan is that rule as a pure function (no device needed): shard `shard` of
 * `n_shards` owns items first, first + stride, ... (count of them). */
int am_shard_plan(size_t n_items, size_t n_shards, size_t shard, size_t* first, size_t* stride, size_t* count);

/* A pool = LibConvolve::new(sample_data) (audio_matcher.rs:289) replicated on each listed
 * device (needle + its spectrum per device, built locally).  devices == NULL: every
 * visible device, in ordinal order.  A device may be listed more than once (its slots then
 * share that device's queue). */
typedef struct am_pool am_pool;
int am_pool_create(const float* needle, size_t n, const int* devices, size_t n_dev, am_pool** out);
void am_pool_destroy(am_pool* pool);
int am_pool_size(const am_pool* pool, size_t* n_dev);
/* slot's device ordinal and its needle handle (borrowed: valid until am_pool_destroy) */
int am_pool_slot(const am_pool* pool, size_t slot, int* device, const am_needle** needle);
/* The whole loop on HOST buffers: one submit thread per slot takes its haystacks in order;
 * a second thread per slot copies haystack i+1 into the other half of a two-slot HBM ring
 * while haystack i is matched, so the link and the kernels overlap.  out / n_out are laid
 * out as in am_match_batch_device (cap_per_hay slots per haystack); every submit thread
 * writes only its own haystacks' slots.  Returns the worst status over all haystacks. */
int am_pool_match_batch(am_pool* pool, const float* const* haystacks, const size_t* lens, size_t n_hay,
                        const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out);
/* Same with resident haystacks: d_haystacks[k] must live on the device of slot k mod n_dev (checked with
 * hipPointerGetAttributes: AM_ERR_INVALID_ARG names the haystack that sits on another device). */
int am_pool_match_batch_device(am_pool* pool, const float* const* d_haystacks, const size_t* lens, size_t n_hay,
                               const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out);

/* The same two loops on interleaved 16-bit stereo PCM (what the reference decodes every file
 * to, mp3_reader.rs:26-37); `frames` per haystack. */
int am_pool_match_batch_pcm16(am_pool* pool, const int16_t* const* interleaved, const size_t* frames, size_t n_hay,
                              const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out);
int am_pool_match_batch_pcm16_device(am_pool* pool, const int16_t* const* d_interleaved, const size_t* frames, size_t n_hay,
                                     const am_match_params* p, am_peak* out, size_t cap_per_hay, size_t* n_out);
/* A pool of SEVERAL equal-length needles (BASELINE config 4 over every GPU of the node): each
 * needle = one LibConvolve::new (audio_matcher.rs:289), replicated per device.  The single-needle
 * calls above refuse such a pool; am_pool_match_multi_batch* run am_match_multi_batch_device per
 * slot on that slot's shard (haystack k on slot k mod n_dev), host buffers through the same
 * two-slot copy ring.  out / n_out: cap_per_pair slots per (haystack, needle) pair at index
 * k * n_needles + j, k the index in the caller's batch. */
int am_pool_create_multi(const float* const* needles, size_t n_needles, size_t n, const int* devices, size_t n_dev, am_pool** out);
int am_pool_needle_count(const am_pool* pool, size_t* n_needles);
int am_pool_match_multi_batch(am_pool* pool, const void* const* haystacks, const size_t* lens, size_t n_hay, int sample_format,
                              const am_match_params* p, am_peak* out, size_t cap_per_pair, size_t* n_out);
int am_pool_match_multi_batch_device(am_pool* pool, const void* const* d_haystacks, const size_t* lens, size_t n_hay, int sample_format,
                                     const am_match_params* p, am_peak* out, size_t cap_per_pair, size_t* n_out);

/* ---- ONE long haystack over several GPUs ------------------------------------------- */
/* calc_chunks fans the windows of ONE haystack out over its workers (iter.par_bridge().map(..),
 * audio_matcher.rs:104-131) and sorts and filters the union afterwards (:132-140).  The same split over
 * devices ("a single very long haystack: shard by chunk ranges with an S-1 halo"): part i of
 * n_parts owns a contiguous range of windows, its samples reach to the end of its last window (chunk +
 * overlap, which contains the S - 1 halo), its windows are matched as on one device and ONE merge runs
 * over all parts, so that a peak next to a cut meets its neighbour from the other part.  The result
 * equals am_match on the whole buffer: offsets and plateau ends identical, heights and prominences to
 * f32 rounding (the overlap-save blocks of a part start at the part, not at sample 0).
 *
 * am_long_plan is the split as a pure function (no device needed): windows [first_window, first_window +
 * n_windows) and samples [first_sample, first_sample + n_samples) of part `part`; n_windows may be 0. */
int am_long_plan(size_t len, size_t needle_len, const am_match_params* p, size_t n_parts, size_t part,
                 size_t* first_window, size_t* n_windows, size_t* first_sample, size_t* n_samples);
/* One part: calc_chunks up to audio_matcher.rs:131 (windowing, correlation, find_peaks, offset restore) on
 * the first n_windows windows of the resident samples d_part[0 .. n_samples); peaks come back UNMERGED, in
 * window order, positions shifted by first_sample.  For hosts that run one process per GPU: every rank
 * matches its part, the ranks' lists are concatenated in part order and am_merge_peaks finishes. */
int am_match_part_device(const am_needle* h, const void* d_part, size_t n_samples, int sample_format, const am_match_params* p,
                         size_t n_windows, uint64_t first_sample, am_peak* out, size_t cap, size_t* n_out);
/* sort by position.start + filter_surrounding(!is_overshadowed) (audio_matcher.rs:132-160) on a host list */
int am_merge_peaks(const am_match_params* p, const am_peak* peaks, size_t n, am_peak* out, size_t cap, size_t* n_out);
/* The three steps over the slots of a single-needle pool, one thread per slot.  Host buffer: every slot
 * copies its own sample range over its own link (the copies of n devices run side by side) and matches
 * it.  _device: d_parts[i] = the samples of part i (am_long_plan with n_parts = the pool's size), resident
 * on the device of slot i.  sample_format: AM_FMT_*; len in samples / frames.  Progress: haystack index 0,
 * chunk indices of the whole haystack. */
int am_pool_match_long(am_pool* pool, const void* haystack, size_t len, int sample_format, const am_match_params* p,
                       am_peak* out, size_t cap, size_t* n_out);
int am_pool_match_long_device(am_pool* pool, const void* const* d_parts, size_t len, int sample_format, const am_match_params* p,
                              am_peak* out, size_t cap, size_t* n_out);

/* ---- progress hooks ---------------------------------------------------------- */
/* The two-stage progress callbacks of calc_chunks (audio_matcher.rs:102-117, 129:
 * f1 when a chunk is picked up, f2 when it is done).
 * Per haystack (aggregate): stage 0 with its chunk count when it is queued, stage 1 when
 * its peaks are back on the host.
 * Per chunk (the reference's granularity): all chunks of a haystack run in one set of
 * launches here, so stage 0 fires for chunks 0..n-1 in order when the haystack is queued
 * and stage 1 for chunks 0..n-1 in order when its results have landed; like the
 * reference's f1/f2, every chunk sees stage 0 before stage 1.
 * haystack_index is the index in the caller's batch (pool calls included; pool submit
 * threads call back concurrently).  Process-wide; a running call keeps the callbacks it
 * started with; pass NULL to clear. */
typedef void (*am_progress_fn)(void* user, size_t haystack_index, int stage, size_t n_chunks);
int am_set_progress_callback(am_progress_fn fn, void* user);
typedef void (*am_chunk_progress_fn)(void* user, size_t haystack_index, size_t chunk_index, size_t n_chunks, int stage);
int am_set_chunk_progress_callback(am_chunk_progress_fn fn, void* user);

/* ---- measurement hooks ---------------------------------------------------- */
/* When enabled every kernel launch of the pipeline on `device` is bracketed by
 * HIP events on the stream it is launched on.  am_profile_query returns the
 * summed elapsed time and launch count of kernels whose name matches `kernel`
 * exactly ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats",
 * "peaks", or "*" for all). */
int am_profile_enable(int device, int on);
int am_profile_reset(int device);
int am_profile_query(int device, const char* kernel, double* total_ms, uint64_t* launches);

/* Measurement hook, not part of the drop-in boundary: the two column kernels on `npairs` block pairs of synthetic
 * input, `iters` launches each, average launch time in ms.  wide = 0: the production 2^22-point plan (512 x 8192);
 * wide = 1: 2^23 points factored 512 x 16384, i.e. the same 512-row kernels on rows twice as long (no row kernel exists
 * for that factorisation yet; DESIGN.md 9.3 sizes it with this).  dense != 0: the inverse kernel writes every score. */
int am_debug_column_bench(int device, int wide, int npairs, int iters, int dense, double* k1_ms, double* k3_ms);

/* Process-wide option DEFAULTS.  A call reads them once on entry, so changing one never
 * affects a call that is already running.
 *   "log_n" (0 = auto), "pairs_per_group", "profile_mask": tuning / measurement knobs
 *   "profile_every" (n >= 1, default 1): with profiling on, bracket only every n-th launch of a kernel class with
 *       events (an event pair costs the stream about 8 us per kernel boundary; am_profile_query then reports the
 *       bracketed launches' time and count)
 *   "batch_overlap" (0/1, default 1): in am_match_batch_device pick the peaks of haystack k
 *       on a second stream beside the transforms of haystack k+1
 *   "needle_group" (1..8, default 8): how many needles of am_match_multi_device share
 *       one forward row transform of the haystack (1 = one row pass per needle)
 *   "half_pipeline" (0/1/2, BASELINE config 5; 0 = off, the default): 1 = the transform's work
 *       matrix travels through HBM in half precision, butterflies stay f32 (scores within about
 *       2e-5 on noise-like audio); 2 = the butterflies run in packed f16 as well (the row kernel and
 *       the forward column kernel whole, the first pass of the inverse column kernel; its
 *       second pass and the score scan stay f32; scores within about 1e-3).  Hit offsets are unaffected.  Meant for audio-level signals: full-scale input
 *       stays inside f16's range, inputs far above full scale may overflow it.
 *   "dense_scores" (0/1): write every raw score from the inverse pass (threshold -inf)
 *       instead of only the tiles that can matter to the peak pick; results are
 *       identical, this is the worst case of the sparse-score path for measurements.
 *   "tail_block" (0/1, default 1): two overlap-save blocks share one complex transform, so a haystack with an odd
 *       number of blocks pays a whole pair for its last, part-filled block.  1 = when the scores behind the last even
 *       block boundary fit one pair of the next smaller transform (half the points), they come from that: beside the
 *       main pass for a single haystack, several haystacks per launch in a batch.  Offsets are unaffected, scores agree
 *       to rounding (1e-6); a haystack's results do not depend on the batch it travels in.  Single-needle entry points
 *       (am_match*, am_pool_match_batch*, am_pool_match_long*, am_match_part_device) and the several-needle engine when
 *       every needle group holds at least two needles (the tail's forward pass once per haystack, its row and inverse
 *       passes once per group); not streaming ingest, not a forced "log_n".
 *   "host_pick_wait" (0/1, default 1): in a batch, the calling thread (not the stream) waits for the peak pick that
 *       last read a set of score buffers before it queues the next haystack into that set -- it runs far ahead of the
 *       GPU either way, and the main stream is spared a barrier packet per haystack (results are identical).
 *   "device_redo" (0/1, default 1): in a batch, chunks whose sparse-score certificate fails get their dense
 *       inverse pass on the device, beside the next haystack's transforms; 0 = the host path does it
 *       after the call's kernels (results are identical; for measurements).
 *   The rules of the path that no source or test available offline pins (the crates find_peaks 0.1 and common are
 *   not in the reference tree).  Defaults (0) = the documented choices of oracle/oracle.c; every
 *   alternative is implemented in the kernels, on the host and in the test checker (same switches), DESIGN.md
 *   section 3 lists inputs on which they differ -- one run of the crates on those pins each rule:
 *     "peak_filter_order"  0: find_peaks filters by prominence, then by distance (audio_matcher.rs:226-229 builder order)
 *                          1: by distance first (every maximum that passes the height test competes), then by prominence:
 *                             scipy.signal.find_peaks' order
 *     "distance_rule"      bit 0: the distance filter drops a peak at a distance  0: <  1: <=  min_distance from a kept, higher one
 *                          bit 1: measured between  0: plateau middles (start + end) / 2   1: plateau starts
 *     "tail_window"        0: chunked(chunk + overlap, hop = chunk) (audio_matcher.rs:104) yields the shorter windows at the
 *                             end of the haystack   1: full-length windows only
 *     "surrounding_from"   filter_surrounding (audio_matcher.rs:136-139): 0: both neighbours from the sorted, unfiltered
 *                             sequence   1: the neighbour before = the last element kept (a sequential filter)
 *   test hooks: "debug_no_realloc" (0/1): a scratch buffer that would have to be (re)allocated while a batch
 *       is being queued fails the call with AM_ERR_HIP instead (every such buffer is sized before the queueing
 *       loop; this makes a violation visible); "debug_redo_arm_at" (k >= 0: the device-side redo of a batch is
 *       armed from haystack k of the call on; -1: never; -2, the default: when a failed certificate has been
 *       seen) -- pins the otherwise timing-dependent choice between the device and the host redo path. */
int am_set_option(const char* key, long long value);
int am_get_option(const char* key, long long* value);
/* "log_n" and "half_pipeline" per needle handle: -1 = follow the process default (initial
 * state), otherwise the handle's own value, which wins over the default. */
int am_needle_set_option(am_needle* h, const char* key, long long value);
int am_needle_get_option(const am_needle* h, const char* key, long long* value);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOMATCH_H */
