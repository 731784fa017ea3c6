"""GPU parity of level 1 (CorrelateAlgo::correlate_with_sample,
audio_matcher.rs:67-72) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: correlation scores within 1e-4 (f32)


def test_k1_reference_known_answer(gpu, oracle):
    """audio_matcher.rs:490-517: within = -10..10, needle [1,2,3], Valid, unscaled -> 6j-52."""
    within = np.arange(-10, 10, dtype=np.float32)
    algo = gpu.HipConvolve([1.0, 2.0, 3.0])
    got = algo.correlate_with_sample(within, gpu.Mode.Valid, False)
    expect = np.array([6 * j - 52 for j in range(18)], dtype=np.float32)
    assert got.shape == expect.shape
    assert np.abs(got - expect).max() < 1.2e-5   # the reference's own bound (audio_matcher.rs:511)


def test_k5_bench_shape(gpu, oracle):
    """benches/my_benchmark.rs:31-32: needle 100..150, haystack -2000..2000, Valid, unscaled."""
    needle = np.arange(100, 150, dtype=np.float32)
    hay = np.arange(-2000, 2000, dtype=np.float32)
    expect = oracle.correlate(hay, needle, oracle.MODE_VALID, oracle.SCALE_NONE, oracle.FFT_DIRECT)
    got = gpu.HipConvolve(needle).correlate_with_sample(hay, gpu.Mode.Valid, False)
    assert got.shape == expect.shape
    rel = np.abs(got - expect).max() / np.abs(expect).max()
    assert rel < 1e-5


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("scale", [0, 1, 2])
@pytest.mark.parametrize("w,s", [(20, 3), (257, 16), (5000, 333), (3, 7), (1, 1), (4097, 4097), (70000, 1025)])
def test_modes_and_scales(gpu, oracle, mode, scale, w, s):
    rng = np.random.default_rng(w * 131 + s)
    within = rng.uniform(-1, 1, w).astype(np.float32)
    needle = rng.uniform(-1, 1, s).astype(np.float32)
    expect = oracle.correlate(within, needle, mode, scale, oracle.FFT_POW2)
    got = gpu.HipConvolve(needle).correlate_with_sample(within, gpu.Mode(mode), gpu.Scale(scale))
    assert got.shape == expect.shape
    ref = max(1.0, float(np.abs(expect).max()))
    assert np.abs(got - expect).max() / ref < TOL


def test_inverse_sample_auto_correlation(gpu, oracle):
    rng = np.random.default_rng(5)
    needle = rng.uniform(-0.25, 0.25, 44100).astype(np.float32)
    got = gpu.HipConvolve(needle).inverse_sample_auto_correlation()
    exp = oracle.inv_autocorr(needle)
    assert abs(got - exp) / exp < 1e-6


def test_multi_block_overlap_save(gpu, oracle):
    """Long within -> several overlap-save blocks and pairs, odd block count."""
    rng = np.random.default_rng(11)
    needle = rng.uniform(-1, 1, 3000).astype(np.float32)
    within = rng.uniform(-1, 1, 300_000).astype(np.float32)
    gpu.set_option("log_n", 14)
    try:
        got = gpu.HipConvolve(needle).correlate_with_sample(within, gpu.Mode.Valid, True)
    finally:
        gpu.set_option("log_n", 0)
    expect = oracle.correlate(within, needle, oracle.MODE_VALID, oracle.SCALE_LIB, oracle.FFT_POW2)
    assert got.shape == expect.shape
    assert np.abs(got - expect).max() < TOL


@pytest.mark.parametrize("log_n", [10, 13, 17, 20, 22, 23])
def test_every_plan_shape(gpu, oracle, log_n):
    """Each transform factorisation N1 x N2 against the oracle."""
    rng = np.random.default_rng(log_n)
    s = 500
    w = min((1 << log_n) + 12345, 1_500_000)
    needle = rng.uniform(-1, 1, s).astype(np.float32)
    within = rng.uniform(-1, 1, w).astype(np.float32)
    gpu.set_option("log_n", log_n)
    try:
        got = gpu.HipConvolve(needle).correlate_with_sample(within, gpu.Mode.Valid, True)
    finally:
        gpu.set_option("log_n", 0)
    expect = oracle.correlate(within, needle, oracle.MODE_VALID, oracle.SCALE_LIB, oracle.FFT_POW2)
    assert np.abs(got - expect).max() < TOL
