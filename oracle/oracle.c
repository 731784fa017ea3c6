/*
 * oracle/oracle.c -- CPU restatement of the reference's matcher hot path.
 * TEST INFRASTRUCTURE ONLY: see the header of oracle/oracle.h for who may use
 * this file and for the pinning status of every function.
 *
 * All file:line citations are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- transforms: one f64 instantiation (checker), one f32 (timing leg) ---- */
#define REAL double
#define SUF d
#include "fft_impl.inc"
#undef REAL
#undef SUF
#define REAL float
#define SUF f
#include "fft_impl.inc"
#undef REAL
#undef SUF

/* ------------------------------------------------------------------------- */
/* mp3_reader.rs:12  const PCM_FACTOR: f32 = 1.0 / ((1 << 16) - 1) as f32     */
/* mp3_reader.rs:35  (*l as f32 + *r as f32) * 0.5 * PCM_FACTOR               */
void orc_pcm_s16_stereo_to_mono(const int16_t* lr, size_t frames, float* out) {
    const float pcm_factor = 1.0f / 65535.0f;
    for (size_t i = 0; i < frames; ++i) {
        float l = (float)lr[2 * i], r = (float)lr[2 * i + 1];
        volatile float sum = l + r;          /* volatile: keep each f32 rounding step */
        volatile float half = sum * 0.5f;
        out[i] = half * pcm_factor;
    }
}

/* audio_matcher.rs:450-456: Full -> whole array; Same -> centered(out, w);
 * Valid -> centered(out, w.saturating_sub(s) + 1) */
size_t orc_mode_len(size_t w, size_t s, int mode) {
    switch (mode) {
    case ORC_MODE_FULL: return w + s - 1;
    case ORC_MODE_SAME: return w;
    default: return (w > s ? w - s : 0) + 1;
    }
}
/* audio_matcher.rs:460-464: start = (arr.len() - len) / 2 */
size_t orc_mode_start(size_t w, size_t s, int mode) {
    size_t full = w + s - 1;
    return (full - orc_mode_len(w, s, mode)) / 2;
}

/* audio_matcher.rs:321-329: 1.0 / correlate(needle, needle, Valid, false)[0].
 * Element 0 of that correlation is sum(needle[n]^2); evaluated in f64. */
float orc_inv_autocorr(const float* needle, size_t s) {
    double acc = 0.0;
    for (size_t i = 0; i < s; ++i) acc += (double)needle[i] * (double)needle[i];
    return (float)(1.0 / acc);
}

static size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

/* spec / spec_pad: an optional cached needle spectrum for pad length spec_pad
 * (ORC_FFT_POW2_CACHED, see orc_calc_chunks) */
static size_t correlate_impl(const float* within, size_t w, const float* needle, size_t s,
                             int mode, int scale, int fft_policy, int precision,
                             float* out, size_t cap, const void* spec, size_t spec_pad) {
    if (w == 0 || s == 0) return (size_t)-1;
    size_t full_len = w + s - 1;
    size_t len = orc_mode_len(w, s, mode), start = orc_mode_start(w, s, mode);
    if (cap < len) return (size_t)-1;
    double* full = (double*)malloc(sizeof(double) * full_len);
    if (!full) return (size_t)-1;
    int rc = 0;
    if (fft_policy == ORC_FFT_DIRECT) {
        /* the definition the transforms implement:
         * full[j] = sum_n within[j + n - (s-1)] * needle[n] */
        for (size_t j = 0; j < full_len; ++j) {
            double acc = 0.0;
            for (size_t n = 0; n < s; ++n) {
                long long idx = (long long)j + (long long)n - (long long)(s - 1);
                if (idx >= 0 && (size_t)idx < w) acc += (double)within[idx] * (double)needle[n];
            }
            full[j] = acc;
        }
    } else {
        size_t pad_len = fft_policy == ORC_FFT_REFERENCE ? full_len : next_pow2(full_len); /* :421 */
        const void* use = (spec && spec_pad == pad_len) ? spec : NULL;
        rc = precision == ORC_PREC_F32
                 ? correlate_full_f(within, w, needle, s, pad_len, full, use)
                 : correlate_full_d(within, w, needle, s, pad_len, full, use);
    }
    if (rc) { free(full); return (size_t)-1; }
    /* scaling: every factor applied as an f32 multiply like scale_slice (:246-252) */
    float factor = 1.0f;
    if (scale == ORC_SCALE_LIB) factor = orc_inv_autocorr(needle, s);                 /* :306-308 */
    else if (scale == ORC_SCALE_MY) factor = orc_inv_autocorr(needle, s) / (float)w;  /* :444-447 */
    for (size_t j = 0; j < len; ++j) {
        float v = (float)full[start + j];
        out[j] = scale == ORC_SCALE_NONE ? v : v * factor;
    }
    free(full);
    return len;
}

size_t orc_correlate(const float* within, size_t w, const float* needle, size_t s,
                     int mode, int scale, int fft_policy, int precision,
                     float* out, size_t cap) {
    return correlate_impl(within, w, needle, s, mode, scale, fft_policy, precision, out, cap, NULL, 0);
}

/* ------------------------------------------------------------------------- */
/* find_peaks 0.1 (crate source absent).  Restated from its documented
 * behaviour, which follows scipy.signal.find_peaks:
 *   1. local maxima incl. flat tops: x[i-1] < x[i], plateau x[i..k) all equal,
 *      x[k] < x[i]; first/last sample are never peaks.  position = i..k.
 *   2. prominence = height - max(left_min, right_min) where each min is taken
 *      walking outwards until a strictly higher sample or the array edge
 *      (pinned by K2: 0.2 / 0.3 / 1.0).
 *   3. keep prominence >= min_prominence.
 *   4. min_distance: greedy by descending height; a peak is dropped when its
 *      middle position is closer than min_distance (strict <) to an already
 *      kept higher peak.                                [PARITY UNPINNED]
 *   5. result ordered by height descending (K2 order: 3, 5, 1).
 * Filter order 3 -> 4 is a choice [PARITY UNPINNED]; with the reference's
 * default min_distance (480 s * sr > chunk length) at most one peak per chunk
 * survives either way -- but WHICH one, or none, differs when the chunk's
 * tallest maximum fails the prominence bound (scipy filters by distance first:
 * the tallest maximum suppresses every other one and is then dropped itself).
 *
 * Every unpinned rule is a field of orc_policy (oracle.h); the defaults are the
 * choices above, the alternatives are restated here and implemented in the
 * library as options of the same names, so that one run by someone who has the
 * crate pins each of them. */
typedef struct { size_t start, end; float height, prom; } pk_t;

static int cmp_height_desc(const void* a, const void* b) {
    const pk_t* x = (const pk_t*)a; const pk_t* y = (const pk_t*)b;
    if (x->height > y->height) return -1;
    if (x->height < y->height) return 1;
    return x->start < y->start ? -1 : (x->start > y->start ? 1 : 0);
}

static size_t dist_pos(const pk_t* p, int from_start) { return from_start ? p->start : (p->start + p->end) / 2; }

size_t orc_find_peaks(const float* y, size_t n, float min_prominence, size_t min_distance,
                      orc_peak* out, size_t cap) {
    return orc_find_peaks_policy(y, n, min_prominence, min_distance, NULL, out, cap);
}

size_t orc_find_peaks_policy(const float* y, size_t n, float min_prominence, size_t min_distance,
                             const orc_policy* pol, orc_peak* out, size_t cap) {
    const int order = pol ? pol->peak_filter_order : 0;
    const int inclusive = pol ? (pol->distance_rule & 1) : 0, from_start = pol ? ((pol->distance_rule >> 1) & 1) : 0;
    if (n < 3) return 0;
    size_t cnt = 0, alloc = 64;
    pk_t* pk = (pk_t*)malloc(sizeof(pk_t) * alloc);
    if (!pk) return 0;
    size_t i = 1, i_max = n - 1;
    while (i < i_max) {
        if (y[i - 1] < y[i]) {
            size_t ahead = i + 1;
            while (ahead < i_max && y[ahead] == y[i]) ++ahead;
            if (y[ahead] < y[i]) {
                if (cnt == alloc) {
                    alloc *= 2;
                    pk_t* np = (pk_t*)realloc(pk, sizeof(pk_t) * alloc);
                    if (!np) { free(pk); return 0; }
                    pk = np;
                }
                pk[cnt].start = i; pk[cnt].end = ahead; pk[cnt].height = y[i]; pk[cnt].prom = 0.f;
                ++cnt;
                i = ahead;
            }
        }
        ++i;
    }
    /* prominence */
    size_t kept = 0;
    for (size_t p = 0; p < cnt; ++p) {
        float h = pk[p].height;
        float lmin = h, rmin = h;
        for (size_t k = pk[p].start; k-- > 0;) { if (y[k] > h) break; if (y[k] < lmin) lmin = y[k]; }
        for (size_t k = pk[p].end; k < n; ++k) { if (y[k] > h) break; if (y[k] < rmin) rmin = y[k]; }
        float base = lmin > rmin ? lmin : rmin;
        pk[p].prom = h - base;
        /* order 0: the prominence filter runs first; order 1: every maximum goes into the distance filter */
        if (order || pk[p].prom >= min_prominence) pk[kept++] = pk[p];
    }
    cnt = kept;
    qsort(pk, cnt, sizeof(pk_t), cmp_height_desc);
    if (min_distance > 0) {
        /* greedy by descending height; the kept list of an order-1 run is quadratic in the number of maxima
         * at worst -- a bucket table (one kept peak per min_distance-wide bucket at most) keeps it linear */
        size_t nb = n / min_distance + 3;
        long long* table = (long long*)malloc(sizeof(long long) * nb);
        if (!table) { free(pk); return 0; }
        for (size_t b = 0; b < nb; ++b) table[b] = -1;
        kept = 0;
        for (size_t p = 0; p < cnt; ++p) {
            size_t mid = dist_pos(&pk[p], from_start);
            size_t bkt = mid / min_distance;
            int ok = 1;
            for (size_t b = bkt > 0 ? bkt - 1 : 0; b <= bkt + 1 && ok; ++b) {
                if (table[b] < 0) continue;
                size_t mq = (size_t)table[b];
                size_t d = mid > mq ? mid - mq : mq - mid;
                if (inclusive ? d <= min_distance : d < min_distance) ok = 0;
            }
            if (ok) { table[bkt] = (long long)mid; pk[kept++] = pk[p]; }
        }
        free(table);
        cnt = kept;
    }
    if (order) {   /* ... and the prominence filter afterwards */
        kept = 0;
        for (size_t p = 0; p < cnt; ++p) if (pk[p].prom >= min_prominence) pk[kept++] = pk[p];
        cnt = kept;
    }
    for (size_t p = 0; p < cnt && p < cap; ++p) {
        out[p].start = pk[p].start; out[p].end = pk[p].end;
        out[p].height = pk[p].height; out[p].prominence = pk[p].prom;
    }
    free(pk);
    return cnt;
}

/* ------------------------------------------------------------------------- */
size_t orc_round_samples(double seconds, uint32_t sr) {
    return (size_t)llround(seconds * (double)sr); /* f64::round = half away from zero */
}

/* Duration::from_secs_f64(start as f64 / sr as f64) (matcher/mod.rs:127-129).
 * std's conversion is exact on the f64 bits with round-to-nearest-even on the
 * nanosecond; done here in 128-bit integers. */
uint64_t orc_start_nanos(uint64_t start, uint32_t sr) {
    double t = (double)start / (double)sr;
    if (!(t > 0.0)) return 0;
    int e;
    double m = frexp(t, &e);                       /* t = m * 2^e, 0.5 <= m < 1 */
    unsigned long long mant = (unsigned long long)ldexp(m, 53); /* 53-bit integer */
    int sh = e - 53;                               /* t = mant * 2^sh */
    unsigned __int128 v = (unsigned __int128)mant * 1000000000ull;
    if (sh >= 0) return (uint64_t)(v << sh);
    int r = -sh;
    if (r >= 127) return 0;
    unsigned __int128 q = v >> r, rem = v & (((unsigned __int128)1 << r) - 1);
    unsigned __int128 half = (unsigned __int128)1 << (r - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    return (uint64_t)q;
}

/* audio_matcher.rs:143-160.  prominence is Option<f32>; it is always Some here
 * because with_min_prominence was set (audio_matcher.rs:227). */
int orc_is_overshadowed(const orc_peak* element, const orc_peak* other, uint32_t sr,
                        double max_distance_s) {
    if (!other) return 0;
    uint64_t e = orc_start_nanos(element->start, sr), b = orc_start_nanos(other->start, sr);
    if (e < b) { uint64_t t = e; e = b; b = t; }
    /* Duration::from_secs(n) / parse_duration produce whole ns; the args path
     * only yields exact values, so the f64 -> ns conversion below is exact */
    uint64_t maxd = (uint64_t)llround(max_distance_s * 1e9);
    return ((e - b) < maxd) && (other->prominence > element->prominence);
}

/* ------------------------------------------------------------------------- */
typedef struct {
    const float* hay; size_t h; const float* needle; size_t s;
    size_t chunk, window; float min_prom; size_t min_dist;
    int scale, policy, prec;
    const orc_policy* pol;
    size_t n_windows;
    const void* spec; size_t spec_pad;   /* cached needle spectrum (ORC_FFT_POW2_CACHED) */
    size_t next;                 /* work counter (par_bridge analogue) */
    pthread_mutex_t mu;
    orc_peak** per_window; size_t* per_window_n;
    int failed;
} cc_job;

static void* cc_worker(void* arg) {
    cc_job* J = (cc_job*)arg;
    for (;;) {
        pthread_mutex_lock(&J->mu);
        size_t i = J->next++;
        pthread_mutex_unlock(&J->mu);
        if (i >= J->n_windows) break;
        size_t off = J->chunk * i;                                  /* :119 */
        size_t w = J->h - off < J->window ? J->h - off : J->window; /* tail window */
        J->per_window[i] = NULL; J->per_window_n[i] = 0;
        if (w < J->s) continue; /* no valid lag; see note in orc_calc_chunks */
        if (J->pol && J->pol->tail_window && w < J->window) continue; /* policy: full-length windows only */
        size_t v = w - J->s + 1;
        float* sc = (float*)malloc(sizeof(float) * v);
        if (!sc) { J->failed = 1; continue; }
        if (correlate_impl(J->hay + off, w, J->needle, J->s, ORC_MODE_VALID, J->scale,
                           J->policy, J->prec, sc, v, J->spec, J->spec_pad) != v) { J->failed = 1; free(sc); continue; } /* :120-122 */
        size_t cap = 16, n;
        orc_peak* pk = (orc_peak*)malloc(sizeof(orc_peak) * cap);
        n = orc_find_peaks_policy(sc, v, J->min_prom, J->min_dist, J->pol, pk, cap);       /* :124 */
        if (n > cap) {
            cap = n; free(pk); pk = (orc_peak*)malloc(sizeof(orc_peak) * cap);
            n = orc_find_peaks_policy(sc, v, J->min_prom, J->min_dist, J->pol, pk, cap);
        }
        for (size_t p = 0; p < n; ++p) { pk[p].start += off; pk[p].end += off; }           /* :126, lib.rs:8-10 */
        J->per_window[i] = pk; J->per_window_n[i] = n;
        free(sc);
    }
    return NULL;
}

static int cmp_start(const void* a, const void* b) {
    const orc_peak* x = (const orc_peak*)a; const orc_peak* y = (const orc_peak*)b;
    return x->start < y->start ? -1 : (x->start > y->start ? 1 : 0);
}

/*
 * calc_chunks (audio_matcher.rs:88-141).
 * Windows: common::chunked(window = chunk+overlap, hop = chunk) -- source absent.
 * Restated as: window i starts at i*chunk while i*chunk < h and holds
 * min(window, h - i*chunk) samples.                          [PARITY UNPINNED]
 * A tail window shorter than the needle has no valid lag; the reference would
 * hand it to fftconvolve (behaviour unknown); it is skipped here. [UNPINNED]
 * Merge: collect in window order, stable sort by position.start (:135),
 * filter_surrounding against the immediate neighbours of the SORTED,
 * UNFILTERED sequence (:136-139).                            [PARITY UNPINNED]
 */
size_t orc_calc_chunks(uint32_t sr, const float* haystack, size_t h,
                       const float* needle, size_t s,
                       size_t chunk, size_t overlap,
                       float min_prominence, size_t min_distance, double overshadow_distance_s,
                       int scale, int fft_policy, int precision, int threads,
                       orc_peak* out, size_t cap) {
    return orc_calc_chunks_policy(sr, haystack, h, needle, s, chunk, overlap, min_prominence, min_distance,
                                  overshadow_distance_s, scale, fft_policy, precision, threads, NULL, out, cap);
}

size_t orc_calc_chunks_policy(uint32_t sr, const float* haystack, size_t h,
                              const float* needle, size_t s,
                              size_t chunk, size_t overlap,
                              float min_prominence, size_t min_distance, double overshadow_distance_s,
                              int scale, int fft_policy, int precision, int threads, const orc_policy* pol,
                              orc_peak* out, size_t cap) {
    if (chunk == 0 || h == 0 || s == 0) return 0;
    cc_job J; memset(&J, 0, sizeof(J));
    J.pol = pol;
    J.hay = haystack; J.h = h; J.needle = needle; J.s = s;
    J.chunk = chunk; J.window = chunk + overlap; J.min_prom = min_prominence; J.min_dist = min_distance;
    J.scale = scale; J.policy = fft_policy; J.prec = precision;
    J.n_windows = (h + chunk - 1) / chunk;
    void* spec = NULL;
    if (fft_policy == ORC_FFT_POW2_CACHED) {
        /* the "good CPU implementation" row of BASELINE.md: power-of-two padding and the
         * needle transformed ONCE for all full-length windows (the reference re-transforms
         * it per chunk, audio_matcher.rs:430) */
        size_t w0 = h < J.window ? h : J.window;
        J.spec_pad = next_pow2(w0 + s - 1);
        spec = precision == ORC_PREC_F32 ? (void*)needle_spectrum_f(needle, s, J.spec_pad)
                                         : (void*)needle_spectrum_d(needle, s, J.spec_pad);
        J.spec = spec;
    }
    J.per_window = (orc_peak**)calloc(J.n_windows, sizeof(orc_peak*));
    J.per_window_n = (size_t*)calloc(J.n_windows, sizeof(size_t));
    pthread_mutex_init(&J.mu, NULL);
    if (threads < 1) threads = 1;
    if ((size_t)threads > J.n_windows) threads = (int)J.n_windows;
    if (threads == 1) cc_worker(&J);
    else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
        for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, cc_worker, &J);
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    size_t total = 0;
    for (size_t i = 0; i < J.n_windows; ++i) total += J.per_window_n[i];
    orc_peak* all = (orc_peak*)malloc(sizeof(orc_peak) * (total ? total : 1));
    size_t k = 0;
    for (size_t i = 0; i < J.n_windows; ++i) {
        for (size_t p = 0; p < J.per_window_n[i]; ++p) all[k++] = J.per_window[i][p];
        free(J.per_window[i]);
    }
    free(J.per_window); free(J.per_window_n);
    free(spec);
    pthread_mutex_destroy(&J.mu);
    /* stable sort by start: insertion-merge via index tiebreak */
    {
        /* qsort is not stable: decorate with the original index in `end`'s
         * high bits is not possible, so do a simple stable merge sort */
        orc_peak* tmp = (orc_peak*)malloc(sizeof(orc_peak) * (total ? total : 1));
        for (size_t width = 1; width < total; width *= 2) {
            for (size_t lo = 0; lo < total; lo += 2 * width) {
                size_t mid = lo + width < total ? lo + width : total;
                size_t hi = lo + 2 * width < total ? lo + 2 * width : total;
                size_t a = lo, b = mid, o = lo;
                while (a < mid && b < hi) tmp[o++] = (cmp_start(&all[b], &all[a]) < 0) ? all[b++] : all[a++];
                while (a < mid) tmp[o++] = all[a++];
                while (b < hi) tmp[o++] = all[b++];
            }
            memcpy(all, tmp, sizeof(orc_peak) * total);
        }
        free(tmp);
    }
    size_t n_out = 0;
    const int from_filtered = pol ? pol->surrounding_from : 0;
    orc_peak last_kept; int have_kept = 0;
    memset(&last_kept, 0, sizeof(last_kept));
    for (size_t i = 0; i < total; ++i) {
        /* policy surrounding_from = 1: the neighbour before is the last element that was kept */
        const orc_peak* before = from_filtered ? (have_kept ? &last_kept : NULL) : (i > 0 ? &all[i - 1] : NULL);
        const orc_peak* after = i + 1 < total ? &all[i + 1] : NULL;
        if (orc_is_overshadowed(&all[i], before, sr, overshadow_distance_s) ||
            orc_is_overshadowed(&all[i], after, sr, overshadow_distance_s))
            continue;
        last_kept = all[i]; have_kept = 1;
        if (n_out < cap) out[n_out] = all[i];
        ++n_out;
    }
    free(all);
    return J.failed ? (size_t)-1 : n_out;
}

/* ------------------------------------------------------------------------- */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
/* SURVEY.md 8d synthetic generator: 24 random bits -> [-1,1) exactly, times amp */
void orc_synth_uniform(uint32_t seed, uint32_t stream, uint64_t first, size_t n, float amp, float* out) {
    uint32_t key = fmix32(seed * 0x9E3779B9u + stream * 0x7F4A7C15u + 0x01234567u);
    for (size_t k = 0; k < n; ++k) {
        uint64_t i = first + k;
        uint32_t hsh = fmix32((uint32_t)i ^ key);
        hsh = fmix32(hsh + stream * 0x9E3779B9u + (uint32_t)(i >> 32) * 0xC2B2AE35u + seed);
        int32_t v = (int32_t)(hsh >> 8) - (1 << 23);
        out[k] = ((float)v * (1.0f / 8388608.0f)) * amp;
    }
}
