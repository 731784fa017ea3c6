// selftest.cpp -- CPU-only checks of the host-side rows (no GPU call): the doctest vectors
// of parse_duration (src/args.rs:66-78), offset formatting and label construction.
#include <cassert>
#include <cstdio>

#include "am_host.hpp"

using namespace amhost;

int main() {
    // src/args.rs:70-78
    assert(parse_duration_ms("17") == 17000u);
    assert(parse_duration_ms("58sec") == 58000u);
    assert(parse_duration_ms("1m") == 60000u);
    assert(parse_duration_ms("100ms") == 100u);
    assert(parse_duration_ms("1hour1m1s") == 3661000u);
    assert(parse_duration_ms("3hours6min1sec") == (3 * 3600 + 6 * 60 + 1) * 1000u);
    assert(parse_duration_ms("3h5m17s") == (3 * 3600 + 5 * 60 + 17) * 1000u);
    assert(!parse_duration_ms(""));
    assert(!parse_duration_ms("3abc"));
    assert(!parse_duration_ms("3s5m"));
    // defaults (matcher/args.rs:19, 70-76)
    const char* argv1[] = {"audiomatch", "a.wav", "b.wav", "--snippet", "s.wav", "--no-out", "-n"};
    Arguments a = parse_arguments(7, argv1);
    assert(a.within.size() == 2 && a.snippet == "s.wav" && a.no_out && a.always_answer == 0);
    assert(a.prominence == 13.0f && a.chunk_size_ms() == 60000 && a.distance_msec() == 480000);
    const am_match_params p = make_params(a, 44100, 10.0);
    assert(p.chunk == 2646000 && p.overlap == 441000 && p.min_distance == 21168000ull);
    assert(std::fabs(p.min_prominence - 0.13f) < 1e-7f && p.scale == AM_SCALE_LIB && p.overshadow_distance_s == 480.0);
    const char* argv2[] = {"audiomatch", "a.wav", "b.wav", "--snippet", "s.wav", "-o", "x.txt"};
    bool threw = false;
    try { parse_arguments(7, argv2); } catch (const ArgError&) { threw = true; }
    assert(threw);   // matcher/mod.rs:20-26
    const char* argv3[] = {"audiomatch", "a.wav", "--snippet", "s.wav", "--distance", "2m", "--chunk-size", "30", "-p", "20"};
    a = parse_arguments(10, argv3);
    assert(a.distance_msec() == 120000 && a.chunk_size_ms() == 30000 && a.prominence == 20.0f);
    // print_offsets (matcher/mod.rs:116-123)
    am_peak pk[3] = {{926100, 926101, 1.0f, 1.0095304f}, {44231234, 44231235, 0.9f, 0.5f}, {158000000, 158000001, 1.f, 1.0f}};
    auto lines = offset_lines(pk, 3, 44100);
    assert(lines[0] == "Offset 1: 00:00:21 with prominence 1.0095304");
    assert(lines[1] == "Offset 2: 00:16:42 with prominence 0.5");
    assert(lines[2] == "Offset 3: 00:59:42 with prominence 1");
    assert(offset_lines(pk, 0, 44100)[0] == "no offsets found");
    // timelabel_from_peaks (archive/data.rs:87-107)
    auto labels = timelabel_from_peaks(pk, 3, 44100, 7.0, "Segment #");
    assert(labels.size() == 2 && labels[0].name == "Segment 1" && labels[1].name == "Segment 2");
    assert(std::fabs(labels[0].start_s - 28.0) < 1e-9 && std::fabs(labels[0].end_s - 44231234.0 / 44100.0) < 1e-9);
    const std::string txt = format_labels(labels);
    assert(txt.rfind("28.000000\t1002.975828\tSegment 1\n", 0) == 0);
    std::printf("am_host selftest ok\n");
    return 0;
}
