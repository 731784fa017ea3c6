// audiomatch_cli.cpp -- `audio-matcher <haystack...> --snippet <needle>` on the GPU library:
// the per-file loop of matcher::run (src/matcher/mod.rs:17-104) with the argument surface of
// src/matcher/args.rs:9-77.  Input files are WAV (PCM16 stereo/mono or float32 mono): MP3
// decoding (minimp3) is outside the accelerated path.
#include <cstdio>
#include <iostream>
#include <sys/stat.h>

#include "am_host.hpp"

using namespace amhost;

static bool file_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

static std::string auto_out_file(const std::string& path) {          // mod.rs:106-108: with_extension("txt")
    const auto slash = path.find_last_of('/');
    const auto dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return path + ".txt";
    return path.substr(0, dot) + ".txt";
}

static bool ask_consent(const Arguments& a, const std::string& question) {   // common::args::input::Inputs
    if (a.always_answer >= 0) return a.always_answer == 1;
    std::fprintf(stderr, "%s [y/n]: ", question.c_str());
    std::string line;
    if (!std::getline(std::cin, line)) return false;
    return !line.empty() && (line[0] == 'y' || line[0] == 'Y' || line[0] == 'j' || line[0] == 'J');
}

static void progress(void*, size_t k, int stage, size_t n_chunks) {          // audio_matcher.rs:102-117, 129
    std::fprintf(stderr, "Progress: file %zu %s (%zu chunks)\n", k, stage == 0 ? "started" : "finished", n_chunks);
}

int main(int argc, char** argv) {
    Arguments args;
    try {
        args = parse_arguments(argc, argv);
    } catch (const ArgError& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    try {
        const Pcm snippet = read_wav(args.snippet);                           // mod.rs:29
        const std::uint32_t sr = snippet.sample_rate;
        const double s_duration = (double)snippet.frames() / (double)sr;      // mod.rs:30 (mp3_duration)
        const std::vector<float> sample_data = to_mono_f32(snippet, args.device);   // mod.rs:32
        am_needle* algo = nullptr;                                            // mod.rs:34: LibConvolve::new
        if (am_needle_create(args.device, sample_data.data(), sample_data.size(), &algo) != AM_OK)
            throw std::runtime_error(std::string("am_needle_create: ") + am_last_error_string());
        if (args.verbosity >= 2) am_set_progress_callback(progress, nullptr);
        int rc_all = 0;
        for (const std::string& main_file : args.within) {                    // mod.rs:42
            std::optional<std::string> out_path = args.out_file;
            if (!out_path && !args.no_out) out_path = auto_out_file(main_file);
            if (out_path && file_exists(*out_path)) {                         // mod.rs:48-66
                if (args.skip_existing ||
                    ask_consent(args, "Ausgabe Datei \"" + *out_path + "\" existiert bereits, möchtest du skippen"))
                    continue;
                if (!ask_consent(args, "soll die existierende Datei überschrieben werden")) out_path.reset();
            }
            if (args.verbosity >= (args.within.size() == 1 ? 3 : 1))
                std::printf("preparing data of '%s'\n", main_file.c_str());
            const Pcm m = read_wav(main_file);                                // mod.rs:71
            if (m.sample_rate != sr) {                                        // mod.rs:72-74 SampleRateMismatch
                std::fprintf(stderr, "sample rate of snippet (%u) and main file (%u) don't match\n", sr, m.sample_rate);
                return 3;
            }
            const std::vector<float> m_samples = to_mono_f32(m, args.device);
            const am_match_params p = make_params(args, sr, s_duration);      // mod.rs:81-87
            std::vector<am_peak> peaks(1024);
            size_t n = 0;
            int rc = am_match(algo, m_samples.data(), m_samples.size(), &p, peaks.data(), peaks.size(), &n);
            if (rc == AM_ERR_CAPACITY) {
                peaks.resize(n);
                rc = am_match(algo, m_samples.data(), m_samples.size(), &p, peaks.data(), peaks.size(), &n);
            }
            if (rc != AM_OK) throw std::runtime_error(std::string("am_match: ") + am_last_error_string());
            if (args.verbosity >= 1)
                for (const auto& line : offset_lines(peaks.data(), n, sr)) std::printf("%s\n", line.c_str());   // mod.rs:89
            if (out_path) {                                                   // mod.rs:92-99
                const std::string text = format_labels(timelabel_from_peaks(peaks.data(), n, sr, 7.0, "Segment #"));
                if (args.dry_run) {
                    std::printf("would write to '%s':\n%s", out_path->c_str(), text.c_str());
                } else {
                    std::ofstream f(*out_path, std::ios::binary | std::ios::trunc);
                    if (!f) { std::fprintf(stderr, "couldn't find file '%s'\n", out_path->c_str()); rc_all = 4; continue; }
                    f << text;
                }
            }
        }
        am_needle_destroy(algo);
        return rc_all;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
