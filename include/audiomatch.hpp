// audiomatch.hpp -- header-only C++17 host mirror of the reference's matcher
// interface (the reference is compiled Rust; no Rust toolchain exists in the
// build image, so the host side above the C ABI is C++):
//
//   trait CorrelateAlgo<f32>          src/matcher/audio_matcher.rs:65-76
//   enum Mode                         src/matcher/audio_matcher.rs:55-59
//   struct Config / PeakConfig        src/matcher/audio_matcher.rs:25-53
//   LibConvolve::new / MyConvolve::new  :289 / :396
//   fn calc_chunks(...)               src/matcher/audio_matcher.rs:88-141
//   find_peaks::Peak<f32>             as consumed at matcher/mod.rs:110-129
//
// Same names, argument meaning and error behaviour: where the trait returns
// Err(Box<dyn Error>) these functions throw audiomatch::Error (calc_chunks in
// the reference unwraps, i.e. panics, audio_matcher.rs:122).  Everything is
// forwarded to libaudiomatch_amd.so (include/audiomatch.h); there is no CPU path.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "audiomatch.h"

namespace audiomatch {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc) {
    if (rc != AM_OK) throw Error(rc, std::string("audiomatch: ") + am_last_error_string());
}

enum class Mode : int { Full = AM_MODE_FULL, Same = AM_MODE_SAME, Valid = AM_MODE_VALID };

// find_peaks::Peak<f32>: position is the half-open range [start, end)
struct Peak {
    std::size_t start = 0, end = 0;
    float height = 0.f;
    float prominence = 0.f;   // Option<f32>; always Some on this path
};

// audio_matcher.rs:25-53, durations in seconds
struct Config {
    double chunk_size = 60.0;        // matcher/args.rs:70-72
    double overlap_length = 0.0;     // Config::from_args: the snippet duration (:41)
    double distance = 8 * 60.0;      // matcher/args.rs:73-76
    float prominence = 13.0f / 100;  // args.prominence / 100 (:44)

    am_match_params params(std::uint32_t sr, bool scale) const {
        am_match_params p{};
        p.sr = sr;
        p.chunk = static_cast<std::uint64_t>(std::llround(chunk_size * sr));       // :100
        p.overlap = static_cast<std::uint64_t>(std::llround(overlap_length * sr)); // :99
        p.min_prominence = prominence;
        p.min_distance = static_cast<std::uint64_t>(distance) * sr;                // distance.as_secs() * sr (:228)
        p.overshadow_distance_s = distance;
        p.scale = scale ? AM_SCALE_LIB : AM_SCALE_NONE;                            // production passes true (mod.rs:85)
        return p;
    }
};

// trait CorrelateAlgo<f32> (audio_matcher.rs:65-76)
class CorrelateAlgo {
public:
    virtual ~CorrelateAlgo() = default;
    virtual float inverse_sample_auto_correlation() const = 0;
    virtual std::vector<float> correlate_with_sample(const float* within, std::size_t len, Mode mode,
                                                     bool scale) const = 0;
    // provided method `scale` (:73-75)
    void scale(std::vector<float>& data) const {
        const float f = inverse_sample_auto_correlation();
        for (float& v : data) v *= f;
    }
};

// The HIP-backed implementation: drop-in for LibConvolve (production, mod.rs:34).
class HipConvolve final : public CorrelateAlgo {
public:
    explicit HipConvolve(const std::vector<float>& sample_data, int device = 0) {
        check(am_needle_create(device, sample_data.data(), sample_data.size(), &h_));
    }
    HipConvolve(const HipConvolve&) = delete;
    HipConvolve& operator=(const HipConvolve&) = delete;
    HipConvolve(HipConvolve&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    ~HipConvolve() override { am_needle_destroy(h_); }

    float inverse_sample_auto_correlation() const override {
        float v = 0.f;
        check(am_needle_inv_autocorr(h_, &v));
        return v;
    }
    std::vector<float> correlate_with_sample(const float* within, std::size_t len, Mode mode,
                                             bool scale) const override {
        std::size_t s = 0, n = 0;
        check(am_needle_len(h_, &s));
        check(am_correlate_len(len, s, static_cast<int>(mode), &n));
        std::vector<float> out(n);
        check(am_correlate(h_, within, len, static_cast<int>(mode), scale ? AM_SCALE_LIB : AM_SCALE_NONE,
                           out.data(), out.size(), &n));
        return out;
    }
    const am_needle* handle() const { return h_; }
    // per-handle "log_n" / "half_pipeline" (-1 = follow the process default)
    void set_option(const char* key, long long value) { check(am_needle_set_option(h_, key, value)); }

private:
    am_needle* h_ = nullptr;
};

// calc_chunks(sr, m_samples, &algo, scale, config) (audio_matcher.rs:88-141):
// peaks sorted by position.start, overshadowed neighbours removed.
inline std::vector<Peak> calc_chunks(std::uint16_t sr, const float* m_samples, std::size_t len,
                                     const HipConvolve& algo_with_sample, bool scale, const Config& config) {
    const am_match_params p = config.params(sr, scale);
    std::vector<am_peak> buf(256);
    std::size_t n = 0;
    int rc = am_match(algo_with_sample.handle(), m_samples, len, &p, buf.data(), buf.size(), &n);
    if (rc == AM_ERR_CAPACITY) {
        buf.resize(n);
        rc = am_match(algo_with_sample.handle(), m_samples, len, &p, buf.data(), buf.size(), &n);
    }
    check(rc);
    std::vector<Peak> out(n);
    for (std::size_t i = 0; i < n; ++i)
        out[i] = Peak{static_cast<std::size_t>(buf[i].start), static_cast<std::size_t>(buf[i].end),
                      buf[i].height, buf[i].prominence};
    return out;
}

// The per-file loop of matcher::run (matcher/mod.rs:42-87) over every GPU of the node: the
// needle replicated per device, haystack k matched on device k mod n (am_pool_*), one submit
// thread per device inside the library, results gathered on the host.
class HipConvolvePool {
public:
    // devices empty: every visible device
    explicit HipConvolvePool(const std::vector<float>& sample_data, const std::vector<int>& devices = {}) {
        check(am_pool_create(sample_data.data(), sample_data.size(), devices.empty() ? nullptr : devices.data(),
                             devices.size(), &p_));
    }
    HipConvolvePool(const HipConvolvePool&) = delete;
    HipConvolvePool& operator=(const HipConvolvePool&) = delete;
    ~HipConvolvePool() { am_pool_destroy(p_); }
    std::size_t size() const {
        std::size_t n = 0;
        check(am_pool_size(p_, &n));
        return n;
    }
    // calc_chunks for every haystack of the batch (host buffers); result k belongs to haystacks[k]
    std::vector<std::vector<Peak>> calc_chunks(std::uint16_t sr, const std::vector<const float*>& haystacks,
                                               const std::vector<std::size_t>& lens, bool scale, const Config& config,
                                               std::size_t cap_per_haystack = 256) {
        const am_match_params p = config.params(sr, scale);
        const std::size_t k = haystacks.size();
        std::vector<am_peak> buf(k * cap_per_haystack);
        std::vector<std::size_t> n(k, 0);
        int rc = am_pool_match_batch(p_, haystacks.data(), lens.data(), k, &p, buf.data(), cap_per_haystack, n.data());
        if (rc == AM_ERR_CAPACITY) {
            for (std::size_t v : n) cap_per_haystack = std::max(cap_per_haystack, v);
            buf.assign(k * cap_per_haystack, am_peak{});
            rc = am_pool_match_batch(p_, haystacks.data(), lens.data(), k, &p, buf.data(), cap_per_haystack, n.data());
        }
        check(rc);
        std::vector<std::vector<Peak>> out(k);
        for (std::size_t i = 0; i < k; ++i)
            for (std::size_t j = 0; j < n[i]; ++j) {
                const am_peak& q = buf[i * cap_per_haystack + j];
                out[i].push_back(Peak{static_cast<std::size_t>(q.start), static_cast<std::size_t>(q.end), q.height, q.prominence});
            }
        return out;
    }

    // the same loop on interleaved i16 stereo frames, the format the reference decodes to
    // (mp3_reader.rs:26-37); lens in frames
    std::vector<std::vector<Peak>> calc_chunks_pcm16(std::uint16_t sr, const std::vector<const std::int16_t*>& haystacks,
                                                     const std::vector<std::size_t>& frames, bool scale, const Config& config,
                                                     std::size_t cap_per_haystack = 256) {
        const am_match_params p = config.params(sr, scale);
        const std::size_t k = haystacks.size();
        std::vector<am_peak> buf(k * cap_per_haystack);
        std::vector<std::size_t> n(k, 0);
        int rc = am_pool_match_batch_pcm16(p_, haystacks.data(), frames.data(), k, &p, buf.data(), cap_per_haystack, n.data());
        if (rc == AM_ERR_CAPACITY) {
            for (std::size_t v : n) cap_per_haystack = std::max(cap_per_haystack, v);
            buf.assign(k * cap_per_haystack, am_peak{});
            rc = am_pool_match_batch_pcm16(p_, haystacks.data(), frames.data(), k, &p, buf.data(), cap_per_haystack, n.data());
        }
        check(rc);
        std::vector<std::vector<Peak>> out(k);
        for (std::size_t i = 0; i < k; ++i)
            for (std::size_t j = 0; j < n[i]; ++j) {
                const am_peak& q = buf[i * cap_per_haystack + j];
                out[i].push_back(Peak{static_cast<std::size_t>(q.start), static_cast<std::size_t>(q.end), q.height, q.prominence});
            }
        return out;
    }

    // calc_chunks on ONE long haystack, its windows split over the pool's devices (audio_matcher.rs:104-131 fans the
    // windows of one haystack out; one sort + overshadow pass over the union, :132-140)
    std::vector<Peak> calc_chunks_long(std::uint16_t sr, const float* haystack, std::size_t len, bool scale, const Config& config,
                                       std::size_t cap = 4096) {
        const am_match_params p = config.params(sr, scale);
        std::vector<am_peak> buf(cap);
        std::size_t n = 0;
        int rc = am_pool_match_long(p_, haystack, len, AM_FMT_F32_MONO, &p, buf.data(), cap, &n);
        if (rc == AM_ERR_CAPACITY) {
            buf.assign(n, am_peak{});
            rc = am_pool_match_long(p_, haystack, len, AM_FMT_F32_MONO, &p, buf.data(), buf.size(), &n);
        }
        check(rc);
        std::vector<Peak> out;
        for (std::size_t j = 0; j < n; ++j)
            out.push_back(Peak{static_cast<std::size_t>(buf[j].start), static_cast<std::size_t>(buf[j].end), buf[j].height, buf[j].prominence});
        return out;
    }

private:
    am_pool* p_ = nullptr;
};

// matcher::run's file loop around SEVERAL snippets (BASELINE config 4) over every GPU: all needles
// replicated per device, the haystack's forward transform shared by the needles of a group.
class HipConvolveMultiPool {
public:
    HipConvolveMultiPool(const std::vector<std::vector<float>>& samples, const std::vector<int>& devices = {}) {
        std::vector<const float*> ptrs;
        for (const auto& v : samples) ptrs.push_back(v.data());
        check(am_pool_create_multi(ptrs.data(), ptrs.size(), samples.empty() ? 0 : samples[0].size(),
                                   devices.empty() ? nullptr : devices.data(), devices.size(), &p_));
        nn_ = samples.size();
    }
    HipConvolveMultiPool(const HipConvolveMultiPool&) = delete;
    HipConvolveMultiPool& operator=(const HipConvolveMultiPool&) = delete;
    ~HipConvolveMultiPool() { am_pool_destroy(p_); }
    // result [k][j]: haystack k against needle j; haystacks are host buffers of `sample_format`
    std::vector<std::vector<std::vector<Peak>>> calc_chunks(std::uint16_t sr, const std::vector<const void*>& haystacks,
                                                            const std::vector<std::size_t>& lens, int sample_format, bool scale,
                                                            const Config& config, std::size_t cap_per_pair = 64) {
        const am_match_params p = config.params(sr, scale);
        const std::size_t k = haystacks.size();
        std::vector<am_peak> buf(k * nn_ * cap_per_pair);
        std::vector<std::size_t> n(k * nn_, 0);
        int rc = am_pool_match_multi_batch(p_, haystacks.data(), lens.data(), k, sample_format, &p, buf.data(), cap_per_pair, n.data());
        if (rc == AM_ERR_CAPACITY) {
            for (std::size_t v : n) cap_per_pair = std::max(cap_per_pair, v);
            buf.assign(k * nn_ * cap_per_pair, am_peak{});
            rc = am_pool_match_multi_batch(p_, haystacks.data(), lens.data(), k, sample_format, &p, buf.data(), cap_per_pair, n.data());
        }
        check(rc);
        std::vector<std::vector<std::vector<Peak>>> out(k, std::vector<std::vector<Peak>>(nn_));
        for (std::size_t i = 0; i < k; ++i)
            for (std::size_t j = 0; j < nn_; ++j)
                for (std::size_t q = 0; q < n[i * nn_ + j]; ++q) {
                    const am_peak& v = buf[(i * nn_ + j) * cap_per_pair + q];
                    out[i][j].push_back(Peak{static_cast<std::size_t>(v.start), static_cast<std::size_t>(v.end), v.height, v.prominence});
                }
        return out;
    }

private:
    am_pool* p_ = nullptr;
    std::size_t nn_ = 0;
};

}  // namespace audiomatch
