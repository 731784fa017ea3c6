"""The C++ host mirror (include/audiomatch.hpp: CorrelateAlgo, HipConvolve, HipConvolvePool, HipConvolveMultiPool, Config, calc_chunks
with the reference's names, audio_matcher.rs:65-141, matcher/mod.rs:42-87) driven from a real C++ program on the GPU."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

SRC = r"""
#include <cstdio>
#include <fstream>
#include <vector>
#include "audiomatch.hpp"
using namespace audiomatch;
static std::vector<float> read_f32(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<float> v((size_t)f.tellg() / 4);
    f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)v.size() * 4);
    return v;
}
int main(int argc, char** argv) {
    const std::vector<float> needle = read_f32(argv[1]), hay = read_f32(argv[2]);
    HipConvolve algo(needle);                                   // LibConvolve::new(sample_data)
    const CorrelateAlgo& trait = algo;                          // used through the trait
    std::printf("inv %.9g\n", trait.inverse_sample_auto_correlation());
    std::vector<float> sc = trait.correlate_with_sample(hay.data(), 5000, Mode::Valid, true);
    std::printf("len %zu first %.9g last %.9g\n", sc.size(), sc.front(), sc.back());
    std::vector<float> raw = trait.correlate_with_sample(hay.data(), 5000, Mode::Same, false);
    trait.scale(raw);                                           // provided method (audio_matcher.rs:73-75)
    std::printf("same %zu %.9g\n", raw.size(), raw[1234]);
    Config cfg; cfg.chunk_size = 10.0; cfg.overlap_length = 1.0; cfg.distance = 5.0; cfg.prominence = 0.13f;
    for (const Peak& p : calc_chunks(8000, hay.data(), hay.size(), algo, true, cfg))
        std::printf("peak %zu %zu %.9g %.9g\n", p.start, p.end, p.height, p.prominence);
    // the file loop of matcher::run over the devices of the node (matcher/mod.rs:42-87)
    HipConvolvePool pool(needle, {0, 0});
    const std::vector<float> half(hay.begin(), hay.begin() + hay.size() / 2);
    const std::vector<std::vector<Peak>> all = pool.calc_chunks(8000, {hay.data(), half.data(), hay.data()},
                                                                {hay.size(), half.size(), hay.size()}, true, cfg);
    std::printf("pool %zu", pool.size());
    for (const auto& one : all) { std::printf(" |"); for (const Peak& p : one) std::printf(" %zu", p.start); }
    std::printf("\n");
    // ONE haystack split over the pool's slots by window ranges (audio_matcher.rs:104-140 fanned out over devices)
    std::printf("long");
    for (const Peak& p : pool.calc_chunks_long(8000, hay.data(), hay.size(), true, cfg)) std::printf(" %zu", p.start);
    std::printf("\n");
    // several snippets over the devices of the node: the second "snippet" is the first one delayed by 100 samples
    std::vector<float> shifted(needle.size(), 0.0f);
    for (size_t i = 100; i < needle.size(); ++i) shifted[i] = needle[i - 100];
    HipConvolveMultiPool multi({needle, shifted}, {0, 0});
    const auto pairs = multi.calc_chunks(8000, {hay.data(), half.data()}, {hay.size(), half.size()}, AM_FMT_F32_MONO, true, cfg);
    std::printf("multi");
    for (const auto& per_hay : pairs) for (const auto& one : per_hay) { std::printf(" |"); for (const Peak& p : one) std::printf(" %zu", p.start); }
    std::printf("\n");
    algo.set_option("log_n", 16);
    std::printf("peaks16 %zu\n", calc_chunks(8000, hay.data(), hay.size(), algo, true, cfg).size());
    try { HipConvolve bad(std::vector<float>{}); } catch (const Error& e) { std::printf("error %d\n", e.code); }
    return 0;
}
"""


def test_cpp_mirror_program(gpu, oracle, tmp_path):
    import build as am_build
    lib = am_build.build_library()
    sr = 8000
    needle = oracle.synth_uniform(51, 0, 0, sr)
    hay = oracle.synth_uniform(51, 1, 0, 30 * sr)
    for t in (4, 17):
        hay[t * sr:t * sr + sr] += needle
    needle.tofile(tmp_path / "needle.f32")
    hay.tofile(tmp_path / "hay.f32")
    (tmp_path / "prog.cpp").write_text(SRC)
    exe = tmp_path / "prog"
    subprocess.check_call([am_build._hipcc(), "-O1", "-std=c++17", "-x", "c++", str(tmp_path / "prog.cpp"), "-o", str(exe),
                           "-I", os.path.join(ROOT, "include"), f"-L{os.path.dirname(lib)}", "-laudiomatch_amd",
                           f"-Wl,-rpath,{os.path.dirname(lib)}"])
    out = subprocess.run([str(exe), str(tmp_path / "needle.f32"), str(tmp_path / "hay.f32")],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    algo = gpu.HipConvolve(needle)
    assert abs(float(out[0].split()[1]) - oracle.inv_autocorr(needle)) < 1e-9
    ref = oracle.correlate(hay[:5000], needle, oracle.MODE_VALID, oracle.SCALE_LIB)
    _, n, _, first, _, last = out[1].split()
    assert int(n) == ref.size and abs(float(first) - ref[0]) < 1e-4 and abs(float(last) - ref[-1]) < 1e-4
    same = oracle.correlate(hay[:5000], needle, oracle.MODE_SAME, oracle.SCALE_LIB)
    assert int(out[2].split()[1]) == 5000 and abs(float(out[2].split()[2]) - same[1234]) < 1e-4
    peaks = [l.split()[1:] for l in out if l.startswith("peak ")]
    exp = oracle.calc_chunks(sr, hay, needle, 10 * sr, sr, 0.13, 5 * sr, 5.0)
    assert [(int(p[0]), int(p[1])) for p in peaks] == [(e[0], e[1]) for e in exp] == [(4 * sr, 4 * sr + 1), (17 * sr, 17 * sr + 1)]
    for p, e in zip(peaks, exp):
        assert abs(float(p[2]) - e[2]) < 1e-4 and abs(float(p[3]) - e[3]) < 1e-4
    pool = [l for l in out if l.startswith("pool")][0]
    assert pool == f"pool 2 | {4 * sr} {17 * sr} | {4 * sr} | {4 * sr} {17 * sr}"
    assert [l for l in out if l.startswith("long")] == [f"long {4 * sr} {17 * sr}"]
    # needle 2 = needle 1 delayed by 100 samples (its first 100 samples zero): its hits lie 100 samples earlier
    multi = [l for l in out if l.startswith("multi")][0]
    assert multi == f"multi | {4 * sr} {17 * sr} | {4 * sr - 100} {17 * sr - 100} | {4 * sr} | {4 * sr - 100}"
    assert [l for l in out if l.startswith("peaks16")] == ["peaks16 2"]
    assert out[-1] == "error 1"          # AM_ERR_INVALID_ARG surfaces as audiomatch::Error
