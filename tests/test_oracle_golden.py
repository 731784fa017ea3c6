"""CPU tests: the oracle against every golden vector the reference's own tests
hold for the path and against independent scipy cross-checks."""
import json
import os

import numpy as np
import pytest

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fixtures.json")))


@pytest.mark.parametrize("fft", [0, 1, 2])
@pytest.mark.parametrize("prec", [0, 1])
def test_k1_correlate_known_answer(oracle, fft, prec):
    k = FIX["K1_correlate_valid_unscaled"]
    got = oracle.correlate(k["within"], k["needle"], oracle.MODE_VALID, oracle.SCALE_NONE, fft, prec)
    tol = k["abs_tol"] if prec == 0 else 2e-5   # the f32 timing leg carries Bluestein's own rounding
    assert np.abs(got - np.array(k["expected"], np.float32)).max() <= tol


def test_k2_find_peaks(oracle):
    k = FIX["K2_find_peaks_prominence_order"]
    pk = oracle.find_peaks(k["y"], k["min_prominence"], 0)
    assert [p[0] for p in pk] == k["expected_starts_in_order"]
    for p, e in zip(pk, k["expected_prominences"]):
        assert abs(p[3] - e) < k["abs_tol"]


def test_k3_overshadow(oracle):
    k = FIX["K3_overshadow_truth_table"]
    pk = oracle.find_peaks(FIX["K2_find_peaks_prominence_order"]["y"], 0.0, 0)
    named = {"p1": pk[0], "p2": pk[1], "p3": pk[2]}
    for c in k["cases"]:
        other = named[c["other"]] if c["other"] else None
        assert oracle.is_overshadowed(named[c["element"]], other, k["sr"], c["distance_s"]) == c["expected"], c


def test_k4_mode_crop(oracle):
    L = oracle.lib()
    for c in FIX["K4_mode_crop"]["cases"]:
        w, s = c["w"], c["s"]
        assert L.orc_mode_len(w, s, oracle.MODE_FULL) == c["full_len"] and L.orc_mode_start(w, s, oracle.MODE_FULL) == 0
        assert L.orc_mode_len(w, s, oracle.MODE_SAME) == c["same"]["len"]
        assert L.orc_mode_start(w, s, oracle.MODE_SAME) == c["same"]["start"]
        assert L.orc_mode_len(w, s, oracle.MODE_VALID) == c["valid"]["len"]
        assert L.orc_mode_start(w, s, oracle.MODE_VALID) == c["valid"]["start"]


@pytest.mark.parametrize("fft", [0, 1, 2])
def test_k5_bench_shape(oracle, fft):
    k = FIX["K5_bench_shape"]
    needle = np.arange(*k["needle_range"], dtype=np.float32)
    hay = np.arange(*k["haystack_range"], dtype=np.float32)
    got = oracle.correlate(hay, needle, oracle.MODE_VALID, oracle.SCALE_NONE, fft)
    exp = np.array(k["expected"], np.float64)
    assert np.abs(got - exp).max() / np.abs(exp).max() < k["rel_tol"]


def test_pcm_downmix_bits(oracle):
    k = FIX["PCM_downmix"]
    got = oracle.pcm_s16_stereo_to_mono(np.array(k["interleaved_lr"], np.int16))
    assert [int(v) for v in got.view(np.uint32)] == k["expected_f32_bits"]


def test_pcm_downmix_in_three_steps_has_the_same_bits(oracle):
    """The kernels form (l + r) as an integer, convert once and multiply once with half the constant
    (am_fft.hip, downmix_s16) where mp3_reader.rs:12, 28-37 converts both, adds and multiplies twice: the same f32
    bits for every sum l + r there is (the checker implements the reference's order)."""
    sums = np.arange(-65536, 65535, dtype=np.int64)
    lr = np.empty((sums.size, 2), np.int16)
    lr[:, 0] = np.clip(sums, -32768, 32767)
    lr[:, 1] = sums - lr[:, 0].astype(np.int64)
    assert np.array_equal(lr.astype(np.int64).sum(axis=1), sums)
    ref = oracle.pcm_s16_stereo_to_mono(np.ascontiguousarray(lr).reshape(-1))
    c = np.float32(0.5) * (np.float32(1.0) / np.float32(65535.0))
    fused = sums.astype(np.float32) * c
    assert fused.dtype == np.float32 and np.array_equal(fused.view(np.uint32), ref.view(np.uint32))


def test_modes_are_crops_of_full(oracle):
    rng = np.random.default_rng(0)
    for w, s in [(20, 3), (64, 64), (100, 37), (5, 9)]:
        a = rng.uniform(-1, 1, w).astype(np.float32)
        b = rng.uniform(-1, 1, s).astype(np.float32)
        full = oracle.correlate(a, b, oracle.MODE_FULL, 0, oracle.FFT_DIRECT)
        for mode in (oracle.MODE_SAME, oracle.MODE_VALID):
            n = oracle.lib().orc_mode_len(w, s, mode)
            st = oracle.lib().orc_mode_start(w, s, mode)
            got = oracle.correlate(a, b, mode, 0, oracle.FFT_DIRECT)
            assert np.array_equal(got, full[st:st + n])


def test_fft_paths_agree_with_scipy(oracle):
    from scipy import signal
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, 3000).astype(np.float32)
    b = rng.uniform(-1, 1, 211).astype(np.float32)
    ref = signal.correlate(a.astype(np.float64), b.astype(np.float64), mode="valid", method="direct")
    for fft in (oracle.FFT_REFERENCE, oracle.FFT_POW2, oracle.FFT_DIRECT):
        got = oracle.correlate(a, b, oracle.MODE_VALID, 0, fft)
        assert np.abs(got - ref).max() < 1e-4
    # scaling: LibConvolve = / sum(needle^2); MyConvolve additionally / within.len()
    e = float(np.sum(b.astype(np.float64) ** 2))
    lib = oracle.correlate(a, b, oracle.MODE_VALID, oracle.SCALE_LIB, oracle.FFT_POW2)
    my = oracle.correlate(a, b, oracle.MODE_VALID, oracle.SCALE_MY, oracle.FFT_POW2)
    assert np.abs(lib - ref / e).max() < 1e-6
    assert np.abs(my - ref / e / a.size).max() < 1e-9
    assert abs(oracle.inv_autocorr(b) - 1.0 / e) / (1.0 / e) < 1e-6


def test_find_peaks_matches_scipy(oracle):
    """Positions, plateau edges and prominences follow scipy's definitions."""
    from scipy import signal
    rng = np.random.default_rng(2)
    y = np.round(rng.standard_normal(4000) * 3).astype(np.float32)   # integers: ties and plateaus, exact in f32/f64
    y[100:104] = 20
    pk = sorted(oracle.find_peaks(y, 2.0, 0, cap=4000))
    idx, props = signal.find_peaks(y.astype(np.float64), prominence=2.0, plateau_size=1)
    assert [p[0] for p in pk] == list(props["left_edges"])
    assert [p[1] - 1 for p in pk] == list(props["right_edges"])
    assert np.array_equal(np.array([p[3] for p in pk]), props["prominences"].astype(np.float32))


def test_find_peaks_order_and_distance(oracle):
    y = np.zeros(200, np.float32)
    for pos, h in [(20, 5.0), (30, 4.0), (45, 6.0), (120, 3.0), (128, 3.5)]:
        y[pos] = h
    pk = oracle.find_peaks(y, 0.5, 0)
    assert [p[2] for p in pk] == sorted([p[2] for p in pk], reverse=True)     # by height, descending
    pk = oracle.find_peaks(y, 0.5, 15)
    assert sorted(p[0] for p in pk) == [20, 45, 128]                           # greedy by height
    pk = oracle.find_peaks(y, 0.5, 1000)
    assert [p[0] for p in pk] == [45]


def test_duration_nanos_and_rounding(oracle):
    assert oracle.start_nanos(44100, 44100) == 1_000_000_000
    assert oracle.start_nanos(1, 3) == 333_333_333
    assert oracle.start_nanos(2, 3) == 666_666_667          # round to nearest
    assert oracle.start_nanos(21_168_000, 44100) == 480_000_000_000
    assert oracle.round_samples(0.5, 3) == 2                # 1.5 -> 2 (half away from zero)
    assert oracle.round_samples(10.0, 44100) == 441000


def test_oracle_regression_vectors(oracle):
    for c in FIX["oracle_regression"]["cases"]:
        s = oracle.round_samples(c["needle_s"], c["sr"])
        h = oracle.round_samples(c["hay_s"], c["sr"])
        needle = oracle.synth_uniform(c["seed"], 0, 0, s)
        hay = oracle.synth_uniform(c["seed"], 1, 0, h)
        for t in c["plants_s"]:
            off = oracle.round_samples(t, c["sr"])
            hay[off:off + s] += needle
        pk = oracle.calc_chunks(c["sr"], hay, needle, oracle.round_samples(c["chunk_s"], c["sr"]), s,
                                c["prominence"], int(c["distance_s"]) * c["sr"], c["distance_s"])
        assert [[p[0], p[1]] for p in pk] == [[e[0], e[1]] for e in c["expected_peaks"]]
        for p, e in zip(pk, c["expected_peaks"]):
            assert abs(p[2] - e[2]) < 1e-6 and abs(p[3] - e[3]) < 1e-6


def test_calc_chunks_threads_and_policies_agree(oracle):
    sr = 8000
    s = sr
    needle = oracle.synth_uniform(5, 0, 0, s)
    hay = oracle.synth_uniform(5, 1, 0, 45 * sr)
    for t in (3, 22, 41):
        hay[t * sr:t * sr + s] += needle
    base = oracle.calc_chunks(sr, hay, needle, 10 * sr, s, 0.13, 5 * sr, 5.0)
    assert [p[0] for p in base] == [3 * sr, 22 * sr, 41 * sr]
    for kw in (dict(threads=4), dict(fft=oracle.FFT_REFERENCE), dict(fft=oracle.FFT_REFERENCE, prec=oracle.PREC_F32)):
        other = oracle.calc_chunks(sr, hay, needle, 10 * sr, s, 0.13, 5 * sr, 5.0, **kw)
        assert [p[0] for p in other] == [p[0] for p in base]
        assert max(abs(a[2] - b[2]) for a, b in zip(other, base)) < 1e-4


def test_overshadow_in_merge(oracle):
    """Two hits 3 s apart in different chunks with distance 25 s: the weaker goes (audio_matcher.rs:136-139)."""
    sr = 8000
    s = 2 * sr
    needle = oracle.synth_uniform(7, 0, 0, s)
    hay = oracle.synth_uniform(7, 1, 0, 70 * sr)
    hay[19 * sr:19 * sr + s] += needle
    hay[22 * sr:22 * sr + s] += 0.5 * needle
    pk = oracle.calc_chunks(sr, hay, needle, 20 * sr, s, 0.13, 25 * sr, 25.0)
    assert [p[0] for p in pk] == [19 * sr]
    pk = oracle.calc_chunks(sr, hay, needle, 20 * sr, s, 0.13, 2 * sr, 2.0)
    assert [p[0] for p in pk] == [19 * sr, 22 * sr]
