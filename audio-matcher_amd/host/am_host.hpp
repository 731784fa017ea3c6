// am_host.hpp -- host-side callers and data formats either side of the hot path
// (SURVEY.md section 8f, rows N1-N4), header-only C++17:
//
//   parse_duration            src/args.rs:80-121 (grammar "3h5m17s", "100ms", bare seconds)
//   Arguments                 src/matcher/args.rs:9-77 (flags, defaults 13 / 60 s / 8 min)
//   read_pcm (WAV)            stands in for mp3_reader::read_mp3 (src/matcher/mp3_reader.rs:13-41):
//                             MP3 decode is out of scope; the i16-stereo down-mix itself runs on
//                             the GPU (am_pcm_s16_stereo_to_mono, mp3_reader.rs:28-37)
//   print_offsets             src/matcher/mod.rs:110-125
//   timelabel_from_peaks      src/archive/data.rs:87-107
//   write_labels              audacity::data::TimeLabel::write (external crate, source absent:
//                             Audacity's label-track text format "start\tend\tname", 6 decimals;
//                             PARITY UNPINNED)
#pragma once

#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <optional>
#include <regex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/audiomatch.h"

namespace amhost {

// ---- src/args.rs:80-121 ------------------------------------------------------
// returns milliseconds; nullopt = Err(NoMatch(arg))
inline std::optional<std::uint64_t> parse_duration_ms(const std::string& arg) {
    if (arg.empty()) return std::nullopt;                                   // :84-87
    bool digits = true;
    for (char c : arg) digits = digits && c >= '0' && c <= '9';
    if (digits) {                                                           // arg.parse::<u64>() -> seconds
        errno = 0;
        const unsigned long long v = std::strtoull(arg.c_str(), nullptr, 10);
        if (errno == 0) return (std::uint64_t)v * 1000u;
    }
    static const std::regex re(
        "^(?:(?:(\\d+)h(?:ours?)?)?(?:(\\d+)m(?:in)?)?(?:(\\d+)s(?:ec)?)?)(?:(\\d+)ms(?:ec)?)?$");
    std::smatch m;
    if (!std::regex_match(arg, m, re)) return std::nullopt;
    std::uint64_t ms = 0;
    if (m[1].matched) ms += std::strtoull(m[1].str().c_str(), nullptr, 10);
    ms *= 60;
    if (m[2].matched) ms += std::strtoull(m[2].str().c_str(), nullptr, 10);
    ms *= 60;
    if (m[3].matched) ms += std::strtoull(m[3].str().c_str(), nullptr, 10);
    ms *= 1000;
    if (m[4].matched) ms += std::strtoull(m[4].str().c_str(), nullptr, 10);
    return ms;
}

// ---- src/matcher/args.rs:9-77 --------------------------------------------------
struct Arguments {
    std::vector<std::string> within;          // positional FILEs
    std::string snippet;                      // --snippet FILE
    float prominence = 13.0f;                 // -p/--prominence, default 13 (:19)
    std::optional<std::uint64_t> distance_ms; // --distance (default 8 min, :73-76)
    std::optional<std::uint64_t> chunk_ms;    // --chunk-size (default 60 s, :70-72)
    bool fancy_bar = false;                   // accepted, no effect here
    bool dry_run = false;
    bool skip_existing = false;
    bool no_out = false;                      // OutFile group (:54-66)
    std::optional<std::string> out_file;
    int always_answer = -1;                   // common::args::input::Inputs: -y -> 1, -n -> 0, else ask
    int verbosity = 1;                        // OutputLevel: --silent 0, default 1 (info), --debug 2, --trace 3
    int device = 0;                           // extension: GPU ordinal

    std::uint64_t chunk_size_ms() const { return chunk_ms.value_or(60ull * 1000); }
    std::uint64_t distance_msec() const { return distance_ms.value_or(8ull * 60 * 1000); }
};

struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };

inline Arguments parse_arguments(int argc, const char* const* argv) {
    Arguments a;
    auto need = [&](int& i) -> std::string {
        if (i + 1 >= argc) throw ArgError(std::string("missing value for ") + argv[i]);
        return argv[++i];
    };
    auto dur = [&](const std::string& v, const char* flag) {
        auto d = parse_duration_ms(v);
        if (!d) throw ArgError(std::string("invalid duration '") + v + "' for " + flag);
        return *d;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        if (s == "--snippet") a.snippet = need(i);
        else if (s == "-p" || s == "--prominence") a.prominence = std::strtof(need(i).c_str(), nullptr);
        else if (s == "--distance") a.distance_ms = dur(need(i), "--distance");
        else if (s == "--chunk-size") a.chunk_ms = dur(need(i), "--chunk-size");
        else if (s == "--fancy-bar") a.fancy_bar = true;
        else if (s == "--dry-run") a.dry_run = true;
        else if (s == "--skip-existing") a.skip_existing = true;
        else if (s == "--no-out") a.no_out = true;
        else if (s == "-o" || s == "--out") a.out_file = need(i);
        else if (s == "-y" || s == "--yes") a.always_answer = 1;
        else if (s == "-n" || s == "--no") a.always_answer = 0;
        else if (s == "--silent" || s == "--quiet") a.verbosity = 0;
        else if (s == "--debug") a.verbosity = 2;
        else if (s == "--trace") a.verbosity = 3;
        else if (s == "--device") a.device = std::atoi(need(i).c_str());
        else if (!s.empty() && s[0] == '-' && s != "-") throw ArgError("unknown option " + s);
        else if (!s.empty()) a.within.push_back(s);
    }
    if (a.snippet.empty()) throw ArgError("--snippet <FILE> is required");
    if (a.no_out && a.out_file) throw ArgError("--no-out and --out are mutually exclusive");      // #[group(multiple = false)]
    if (a.out_file && a.within.size() != 1)
        throw ArgError("providet outfile only compatible with one main file");                     // matcher/mod.rs:20-26
    return a;
}

// ---- PCM input (WAV) -------------------------------------------------------------
struct Pcm {
    std::uint32_t sample_rate = 0;
    std::uint16_t channels = 0;
    bool is_float = false;
    std::vector<std::int16_t> s16;   // interleaved, when !is_float
    std::vector<float> f32;          // mono float, when is_float
    std::size_t frames() const { return is_float ? f32.size() : (channels ? s16.size() / channels : 0); }
};

inline Pcm read_wav(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("couldn't find file '" + path + "'");     // CliError::NoFile
    auto rd = [&](void* p, std::size_t n) { f.read(static_cast<char*>(p), (std::streamsize)n); return (std::size_t)f.gcount() == n; };
    char riff[12];
    if (!rd(riff, 12) || std::memcmp(riff, "RIFF", 4) || std::memcmp(riff + 8, "WAVE", 4))
        throw std::runtime_error("'" + path + "' is not a RIFF/WAVE file");
    Pcm pcm;
    std::uint16_t fmt = 0, bits = 0;
    bool have_fmt = false;
    for (;;) {
        char id[4]; std::uint32_t sz = 0;
        if (!rd(id, 4) || !rd(&sz, 4)) break;
        if (!std::memcmp(id, "fmt ", 4)) {
            std::vector<char> b(sz);
            if (!rd(b.data(), sz) || sz < 16) throw std::runtime_error("bad fmt chunk");
            std::memcpy(&fmt, &b[0], 2); std::memcpy(&pcm.channels, &b[2], 2);
            std::memcpy(&pcm.sample_rate, &b[4], 4); std::memcpy(&bits, &b[14], 2);
            if (fmt == 0xFFFE && sz >= 26) std::memcpy(&fmt, &b[24], 2);   // WAVE_FORMAT_EXTENSIBLE sub-format
            have_fmt = true;
        } else if (!std::memcmp(id, "data", 4)) {
            if (!have_fmt) throw std::runtime_error("data chunk before fmt chunk");
            if (fmt == 1 && bits == 16 && (pcm.channels == 1 || pcm.channels == 2)) {
                pcm.s16.resize(sz / 2);
                rd(pcm.s16.data(), (sz / 2) * 2);
            } else if (fmt == 3 && bits == 32 && pcm.channels == 1) {
                pcm.is_float = true;
                pcm.f32.resize(sz / 4);
                rd(pcm.f32.data(), (sz / 4) * 4);
            } else {
                throw std::runtime_error("unsupported WAV encoding (need PCM16 mono/stereo or float32 mono)");
            }
            return pcm;
        } else {
            f.seekg(sz + (sz & 1), std::ios::cur);
        }
    }
    throw std::runtime_error("'" + path + "' has no data chunk");
}

// f32 mono samples as the matcher sees them: stereo i16 goes through the GPU
// down-mix (mp3_reader.rs:28-37); mono i16 is treated as l == r.
inline std::vector<float> to_mono_f32(const Pcm& pcm, int device) {
    if (pcm.is_float) return pcm.f32;
    std::vector<std::int16_t> stereo;
    const std::int16_t* src = pcm.s16.data();
    std::size_t frames = pcm.frames();
    if (pcm.channels == 1) {
        stereo.resize(frames * 2);
        for (std::size_t i = 0; i < frames; ++i) stereo[2 * i] = stereo[2 * i + 1] = pcm.s16[i];
        src = stereo.data();
    }
    std::vector<float> out(frames);
    if (frames) {
        const int rc = am_pcm_s16_stereo_to_mono(device, src, frames, out.data());
        if (rc != AM_OK) throw std::runtime_error(std::string("am_pcm_s16_stereo_to_mono: ") + am_last_error_string());
    }
    return out;
}

// ---- output ------------------------------------------------------------------------
// Rust's `{}` for f32: the shortest decimal that round-trips
inline std::string fmt_f32(float v) {
    char buf[64];
    for (int p = 1; p <= 9; ++p) {
        std::snprintf(buf, sizeof buf, "%.*g", p, (double)v);
        if (std::strtof(buf, nullptr) == v) break;
    }
    std::string s = buf;
    const auto e = s.find('e');
    if (e != std::string::npos) {   // Rust never prints an exponent for Display: expand it
        std::snprintf(buf, sizeof buf, "%.*f", 50, (double)v);
        s = buf;
        while (!s.empty() && s.back() == '0') s.pop_back();
        if (!s.empty() && s.back() == '.') s.pop_back();
    }
    return s;
}

// matcher/mod.rs:110-125: "Offset i: hh:mm:ss with prominence p"
inline std::vector<std::string> offset_lines(const am_peak* peaks, std::size_t n, std::uint32_t sr) {
    std::vector<std::string> out;
    if (n == 0) { out.push_back("no offsets found"); return out; }
    for (std::size_t i = 0; i < n; ++i) {
        const double secs = (double)peaks[i].start / (double)sr;     // start_as_duration (:127-129)
        const std::uint64_t whole = (std::uint64_t)secs;
        char buf[160];
        std::snprintf(buf, sizeof buf, "Offset %zu: %02" PRIu64 ":%02" PRIu64 ":%02" PRIu64 " with prominence %s", i + 1,
                      whole / 3600, (whole / 60) % 60, whole % 60, fmt_f32(peaks[i].prominence).c_str());
        out.push_back(buf);
    }
    return out;
}

struct TimeLabel { double start_s, end_s; std::string name; };

// archive/data.rs:87-107: consecutive peak pairs -> [start_i + delay, start_{i+1}], name_pattern with '#' -> i (from 1)
inline std::vector<TimeLabel> timelabel_from_peaks(const am_peak* peaks, std::size_t n, std::uint32_t sr,
                                                   double delay_start_s, const std::string& name_pattern) {
    std::vector<TimeLabel> out;
    for (std::size_t i = 0; i + 1 < n; ++i) {
        TimeLabel t;
        t.start_s = (double)peaks[i].start / (double)sr + delay_start_s;
        t.end_s = (double)peaks[i + 1].start / (double)sr;
        t.name = name_pattern;
        std::string num = std::to_string(i + 1), rep;
        for (char c : t.name) { if (c == '#') rep += num; else rep += c; }
        t.name = rep;
        out.push_back(t);
    }
    return out;
}

inline std::string format_labels(const std::vector<TimeLabel>& labels) {
    std::string s;
    char buf[128];
    for (const auto& l : labels) {
        std::snprintf(buf, sizeof buf, "%.6f\t%.6f\t", l.start_s, l.end_s);
        s += buf; s += l.name; s += "\n";
    }
    return s;
}

// (secs * sr).round() as usize (audio_matcher.rs:99-100)
inline std::uint64_t round_samples_ms(std::uint64_t ms, std::uint32_t sr) {
    return (std::uint64_t)std::llround((double)ms / 1000.0 * (double)sr);
}

// Config::from_args (audio_matcher.rs:38-52) + the rounding of :99-100, :228
inline am_match_params make_params(const Arguments& a, std::uint32_t sr, double snippet_duration_s) {
    am_match_params p{};
    p.sr = sr;
    p.chunk = round_samples_ms(a.chunk_size_ms(), sr);
    p.overlap = (std::uint64_t)std::llround(snippet_duration_s * (double)sr);
    p.min_prominence = a.prominence / 100.0f;
    p.min_distance = (a.distance_msec() / 1000) * (std::uint64_t)sr;     // distance.as_secs() as usize * sr
    p.overshadow_distance_s = (double)a.distance_msec() / 1000.0;
    p.scale = AM_SCALE_LIB;                                              // matcher/mod.rs:85 passes true
    return p;
}

}  // namespace amhost
