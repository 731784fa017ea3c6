"""GPU parity of the peak pick (find_peaks, audio_matcher.rs:221-230) and the
overshadow filter against the reference's own known answers and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def as_tuples(peaks):
    return [(p.start, p.end, p.height, p.prominence) for p in peaks]


def test_k2_reference_prominences_and_order(gpu):
    """audio_matcher.rs:167-185: peaks of [0,.7,.5,1,.5,.8,0], min prominence 0
    -> start 3 (prom 1.0), start 5 (prom 0.3), start 1 (prom 0.2), in that order."""
    pk = gpu.find_peaks([0.0, 0.7, 0.5, 1.0, 0.5, 0.8, 0.0], 0.0)
    assert [p.start for p in pk] == [3, 5, 1]
    for p, e in zip(pk, [1.0, 0.3, 0.2]):
        assert abs(p.prominence - e) < 1e-6   # tolerance of the reference test


@pytest.mark.parametrize("n,seed", [(7, 0), (100, 1), (1023, 2), (1024, 3), (1025, 4), (5000, 5),
                                    (70001, 6), (300000, 7)])
@pytest.mark.parametrize("min_prom", [0.0, 0.3, 1.5])
def test_matches_oracle_bit_exact(gpu, oracle, n, seed, min_prom):
    """Integer positions, heights and prominences are exact functions of the
    f32 score array: the GPU must agree bit for bit (plateaus included)."""
    rng = np.random.default_rng(seed)
    y = rng.standard_normal(n).astype(np.float32)
    y = np.round(y * 4) / 4  # coarse grid -> many flat tops and ties
    y = y.astype(np.float32)
    exp = oracle.find_peaks(y, min_prom, 0, cap=n)
    # (lists longer than AM_MAX_PEAKS_PER_CHUNK -- up to 56 000 peaks here -- take the global-memory
    # sort of peaks_big_finish; find_peaks returns them all, so does the library)
    got = as_tuples(gpu.find_peaks(y, min_prom, cap=n))
    assert got == exp


@pytest.mark.parametrize("min_dist", [1, 10, 100, 5000])
def test_min_distance(gpu, oracle, min_dist):
    rng = np.random.default_rng(42)
    y = np.cumsum(rng.standard_normal(20000)).astype(np.float32)
    y -= np.linspace(y[0], y[-1], y.size).astype(np.float32)
    exp = oracle.find_peaks(y, 2.0, min_dist, cap=y.size)
    got = as_tuples(gpu.find_peaks(y, 2.0, min_dist))
    assert got == exp


def test_edges_and_degenerate(gpu, oracle):
    for y in ([1.0], [1.0, 2.0], [1.0, 2.0, 1.0], [2.0, 1.0, 2.0], [1.0, 1.0, 1.0, 1.0],
              [0.0, 1.0, 1.0, 1.0, 0.0], [0.0, 1.0, 1.0, 1.0], [1.0, 1.0, 0.0], [0, 1, 0, 1, 0, 1, 0]):
        y = np.asarray(y, dtype=np.float32)
        assert as_tuples(gpu.find_peaks(y, 0.0)) == oracle.find_peaks(y, 0.0, 0)


def test_long_plateau_across_tiles(gpu, oracle):
    y = np.zeros(5000, dtype=np.float32)
    y[900:3100] = 1.0          # flat top spanning three 1024-tiles
    y[4000] = 0.5
    assert as_tuples(gpu.find_peaks(y, 0.1)) == oracle.find_peaks(y, 0.1, 0)


def test_far_walks_use_tile_summaries(gpu, oracle):
    """Two tall peaks far apart: prominence walks cross hundreds of tiles."""
    rng = np.random.default_rng(3)
    y = (rng.standard_normal(1_000_000) * 0.01).astype(np.float32)
    y[123_456] = 1.0
    y[876_543] = 0.9
    y[876_600] = 0.95
    exp = oracle.find_peaks(y, 0.13, 0, cap=100)
    got = as_tuples(gpu.find_peaks(y, 0.13))
    assert got == exp and len(got) == 3


@pytest.mark.parametrize("n", [1500, 1537, 1800, 2046, 2047, 2049, 3000])
def test_chunks_between_one_and_two_tiles(gpu, oracle, n):
    """Score arrays of 1.5 .. 3 k samples: with or without a full 1024-score tile inside, the
    head / tail pieces are longer than the LDS window and go in slices."""
    rng = np.random.default_rng(n)
    y = (rng.uniform(-1, 1, n) + 0.8 * np.sin(np.arange(n) / 37.0)).astype(np.float32)
    for prom, dist in ((0.0, 0), (0.3, 0), (0.3, 25), (0.9, 5000)):
        exp = oracle.find_peaks(y, prom, dist, cap=4096)
        if len(exp) > 1000:
            continue
        got = gpu.find_peaks(y, prom, dist)
        assert [(g.start, g.end, g.height, g.prominence) for g in got] == [tuple(e) for e in exp], (n, prom, dist)
