#!/usr/bin/env python3
"""Writes tests/golden/*.json.

The reference (a Rust crate) cannot be built or run in this environment, so
these fixtures are NOT outputs of the reference.  They are:
  * the inputs, tolerances and expected values that the reference's own tests
    hold (transcribed as data, with file:line), and
  * answers obtained from the mathematical definition by exact direct summation
    in float64 (integer-valued inputs, so the sums are exact),
and a few seeded end-to-end vectors produced by the oracle itself, which pin the
oracle against accidental change (they are marked "oracle_regression").
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))


def direct_valid(within, needle):
    w, s = len(within), len(needle)
    return [float(sum(np.float64(within[j + n]) * np.float64(needle[n]) for n in range(s)))
            for j in range(w - s + 1)]


def main():
    out = {}
    # K1 -- audio_matcher.rs:490-517 my_correlate_same_fftcorrelate
    within = [float(v) for v in range(-10, 10)]
    needle = [1.0, 2.0, 3.0]
    out["K1_correlate_valid_unscaled"] = {
        "source": "src/matcher/audio_matcher.rs:490-517 (inputs, mode Valid, scale=false, abs tolerance 1.2e-5 between "
                  "MyConvolve and LibConvolve); expected values by direct summation out[j] = 6j - 52",
        "within": within, "needle": needle, "mode": "Valid", "scale": False,
        "expected": direct_valid(within, needle), "abs_tol": 1.2e-5}
    assert out["K1_correlate_valid_unscaled"]["expected"] == [6.0 * j - 52 for j in range(18)]
    # K2 -- audio_matcher.rs:167-185
    out["K2_find_peaks_prominence_order"] = {
        "source": "src/matcher/audio_matcher.rs:167-185 overshadow_tests::test_data",
        "y": [0.0, 0.7, 0.5, 1.0, 0.5, 0.8, 0.0], "min_prominence": 0.0,
        "expected_starts_in_order": [3, 5, 1], "expected_prominences": [1.0, 0.3, 0.2], "abs_tol": 1e-6}
    # K3 -- audio_matcher.rs:187-218
    out["K3_overshadow_truth_table"] = {
        "source": "src/matcher/audio_matcher.rs:187-218 (sr = 1; p1 = start 3, p2 = start 5, p3 = start 1)",
        "sr": 1,
        "cases": [
            {"element": "p3", "other": "p1", "distance_s": 3, "expected": True},
            {"element": "p3", "other": "p1", "distance_s": 2, "expected": False},
            {"element": "p2", "other": "p1", "distance_s": 3, "expected": True},
            {"element": "p2", "other": "p1", "distance_s": 2, "expected": False},
            {"element": "p1", "other": None, "distance_s": 6, "expected": False},
            {"element": "p2", "other": None, "distance_s": 6, "expected": False},
            {"element": "p3", "other": None, "distance_s": 6, "expected": False},
            {"element": "p1", "other": "p2", "distance_s": 6, "expected": False},
            {"element": "p1", "other": "p3", "distance_s": 6, "expected": False}]}
    # K4 -- audio_matcher.rs:450-464 crop offsets
    out["K4_mode_crop"] = {
        "source": "src/matcher/audio_matcher.rs:450-464: Full = whole array, Same = centered(out, w), "
                  "Valid = centered(out, w.saturating_sub(s) + 1), start = (len(out) - len) / 2",
        "cases": [{"w": w, "s": s,
                   "full_len": w + s - 1,
                   "same": {"len": w, "start": (s - 1) // 2},
                   "valid": {"len": max(w - s, 0) + 1, "start": (w + s - 1 - (max(w - s, 0) + 1)) // 2}}
                  for (w, s) in [(20, 3), (4000, 50), (3087000, 441000), (5, 5), (3, 7), (1, 1), (10, 4)]]}
    # K5 -- benches/my_benchmark.rs:31-32
    needle5 = [float(v) for v in range(100, 150)]
    hay5 = [float(v) for v in range(-2000, 2000)]
    out["K5_bench_shape"] = {
        "source": "benches/my_benchmark.rs:31-32 (needle 100..150, haystack -2000..2000, Mode::Valid, unscaled); "
                  "expected by exact direct summation",
        "needle_range": [100, 150], "haystack_range": [-2000, 2000],
        "expected": direct_valid(hay5, needle5), "rel_tol": 1e-5}
    # PCM down-mix -- mp3_reader.rs:12, 28-37
    lr = [32767, 32767, -32768, -32768, 32767, -32768, 0, 1, 1, 1, -1, 0, 12345, -23456]
    pcm = np.float32(1.0) / np.float32(65535.0)
    exp = []
    for i in range(0, len(lr), 2):
        v = np.float32(np.float32(np.float32(lr[i]) + np.float32(lr[i + 1])) * np.float32(0.5)) * pcm
        exp.append(int(np.float32(v).view(np.uint32)))
    out["PCM_downmix"] = {
        "source": "src/matcher/mp3_reader.rs:12, 28-37: (l as f32 + r as f32) * 0.5 * (1.0 / 65535 as f32), every step in f32",
        "interleaved_lr": lr, "expected_f32_bits": exp}
    # defaults -- matcher/args.rs:19, 70-76, audio_matcher.rs:44
    out["defaults"] = {
        "source": "src/matcher/args.rs:19 (prominence 13.0), :70-72 (chunk 60 s), :73-76 (distance 8 min); "
                  "audio_matcher.rs:44 (prominence / 100), :228 (distance.as_secs() * sr)",
        "prominence": 0.13, "chunk_s": 60, "distance_s": 480, "min_distance_samples_at_44100": 480 * 44100}
    # oracle regression vectors (NOT reference outputs)
    import pyoracle as po
    reg = []
    for (seed, sr, needle_s, hay_s, plants, chunk_s, dist_s) in [
            (1, 8000, 1.0, 30.0, [4.0, 17.0], 10.0, 5.0),
            (7, 8000, 2.0, 70.0, [5.0, 31.0, 64.5], 20.0, 25.0),
            (3, 4000, 0.5, 12.0, [], 5.0, 480.0)]:
        s = po.round_samples(needle_s, sr)
        h = po.round_samples(hay_s, sr)
        needle = po.synth_uniform(seed, 0, 0, s)
        hay = po.synth_uniform(seed, 1, 0, h)
        for t in plants:
            off = po.round_samples(t, sr)
            hay[off:off + s] += needle
        pk = po.calc_chunks(sr, hay, needle, po.round_samples(chunk_s, sr), s, 0.13, int(dist_s) * sr, dist_s)
        reg.append({"seed": seed, "sr": sr, "needle_s": needle_s, "hay_s": hay_s, "plants_s": plants,
                    "chunk_s": chunk_s, "distance_s": dist_s, "prominence": 0.13,
                    "expected_peaks": [[p[0], p[1], float(np.float32(p[2])), float(np.float32(p[3]))] for p in pk]})
    out["oracle_regression"] = {"source": "oracle/oracle.c itself (pins the oracle, not the reference)", "cases": reg}
    with open(os.path.join(HERE, "fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "fixtures.json"))


if __name__ == "__main__":
    main()
