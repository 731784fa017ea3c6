"""Full-size (BASELINE configs[1]: 10 s needle vs 1 h haystack, 44.1 kHz) checks through
size-independent properties, since the oracle needs minutes at this size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 44100
S = 10 * SR
H = 3600 * SR


def plant_offsets(k):
    return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]


@pytest.fixture(scope="module")
def full(gpu):
    needle = gpu.synth_uniform_device(0, S, seed=1, stream=0)
    algo = gpu.HipConvolve.from_device(0, needle.ptr, S)
    hay = gpu.synth_uniform_device(0, H, seed=1, stream=1)
    for t in plant_offsets(0):
        gpu.axpy_device(0, hay, t, needle.ptr, S, 1.0)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=480.0, prominence=0.13)
    return gpu, needle, algo, hay, cfg.params(SR, gpu.Scale.LIB)


def test_planted_offsets_recovered_and_idempotent(full):
    gpu, needle, algo, hay, p = full
    first = algo.match_device(hay.ptr, H, p)       # first call: every score written
    second = algo.match_device(hay.ptr, H, p)      # later calls: sparse scores
    third = algo.match_device(hay.ptr, H, p)
    assert [q.start for q in first] == plant_offsets(0)
    key = lambda r: [(q.start, q.end, q.height, q.prominence) for q in r]
    assert key(first) == key(second) == key(third)  # bitwise idempotent across the two score paths
    for q in first:
        assert abs(q.height - 1.0) < 0.02 and q.prominence > 0.9


def test_shift_equivariance(full):
    """Matching a haystack that starts 12 345 samples later moves every offset by -12 345."""
    gpu, needle, algo, hay, p = full
    k = 12345
    shifted = algo.match_device(hay.ptr + 4 * k, H - k, p)
    assert [q.start for q in shifted] == [t - k for t in plant_offsets(0)]


def test_score_checksum_full_vector(full, oracle):
    """Sum of all H-S+1 scores against the closed form sum_n needle[n]*(P[n+J]-P[n]) / E,
    P = prefix sums of the haystack (f64 on the host)."""
    gpu, needle, algo, hay, p = full
    J = H - S + 1
    out = gpu.DeviceBuffer(0, 4 * J)
    n = __import__("ctypes").c_size_t(0)
    gpu._check(gpu.lib().am_correlate_device(algo._h, hay.ptr, H, int(gpu.Mode.Valid), int(gpu.Scale.LIB),
                                            out.ptr, J, __import__("ctypes").byref(n)))
    assert n.value == J
    scores = out.to_numpy(np.float32, J)
    h_host = hay.to_numpy(np.float32, H)
    n_host = needle.to_numpy(np.float32, S)
    P = np.concatenate(([0.0], np.cumsum(h_host, dtype=np.float64)))
    e = float(np.sum(n_host.astype(np.float64) ** 2))
    idx = np.arange(S)
    expect = float(np.sum(n_host.astype(np.float64) * (P[idx + J] - P[idx]))) / e
    got = float(np.sum(scores, dtype=np.float64))
    # tolerance: J * per-score error budget (1e-6) is far above f32 accumulation noise here
    assert abs(got - expect) < 1e-6 * J
    # weighted checksum (alternating signs) catches permutations that a plain sum cannot
    w = np.where(np.arange(J) & 1, -1.0, 1.0)
    alt = np.concatenate(([0.0], np.cumsum(h_host.astype(np.float64) * np.where(np.arange(H) & 1, -1.0, 1.0))))
    expect_alt = float(np.sum(n_host.astype(np.float64) * np.where(idx & 1, -1.0, 1.0) * (alt[idx + J] - alt[idx]))) / e
    got_alt = float(np.sum(scores.astype(np.float64) * w))
    assert abs(got_alt - expect_alt) < 1e-6 * J
    # the needle energy behind every scaled score (audio_matcher.rs:321-329)
    assert abs(algo.inverse_sample_auto_correlation() * e - 1.0) < 1e-6
    # spot values against the direct definition: both ends, block interiors, every plant
    n64 = n_host.astype(np.float64)
    for j in (0, 1, 777_777, 1_655_807, 1_655_808, J - 2, J - 1, *plant_offsets(0)):
        direct = float(np.dot(h_host[j:j + S].astype(np.float64), n64)) / e
        assert abs(float(scores[j]) - direct) < 1e-4, j


def test_needle_from_device_memory_is_complete(full):
    """Regression: the needle handle copies a device buffer the caller may have just
    written; its energy (and with it every scaled score) must not depend on timing.
    A fresh source buffer per round keeps the copy on the cold path."""
    gpu, needle, algo, hay, p = full
    n = 4_000_000
    src_host = np.random.default_rng(3).uniform(-0.25, 0.25, n).astype(np.float32)
    expect = 1.0 / float(np.sum(src_host.astype(np.float64) ** 2))
    got = set()
    for _ in range(20):
        src = gpu.DeviceBuffer.from_numpy(0, src_host)
        a = gpu.HipConvolve.from_device(0, src.ptr, n)
        got.add(a.inverse_sample_auto_correlation())
        a.close()
        src.free()
    assert len(got) == 1                      # bitwise reproducible
    assert abs(got.pop() / expect - 1.0) < 1e-6


def test_batch_equals_singles_full_size(full):
    gpu, needle, algo, hay, p = full
    hay2 = gpu.synth_uniform_device(0, H, seed=1, stream=2)
    for t in plant_offsets(1):
        gpu.axpy_device(0, hay2, t, needle.ptr, S, 1.0)
    res = algo.match_batch_device([hay.ptr, hay2.ptr, hay.ptr], [H, H, H], p)
    assert [q.start for q in res[0]] == plant_offsets(0)
    assert [q.start for q in res[1]] == plant_offsets(1)
    assert [(q.start, q.height) for q in res[2]] == [(q.start, q.height) for q in res[0]]


def test_config4_32_needles_full_size(gpu):
    """BASELINE configs[3] at its stated shape on one GPU: 32 needles (streams 2001..2032,
    SURVEY.md 8d) against one 1 h haystack, each needle planted twice; the shared forward
    pass + grouped row transforms (4 groups of 8) find exactly the planted offsets, on the
    dense and on the sparse score path, and agree with separate single-needle calls."""
    nn = 32
    needles = [gpu.synth_uniform_device(0, S, seed=1, stream=2001 + j) for j in range(nn)]
    algos = [gpu.HipConvolve.from_device(0, n.ptr, S) for n in needles]
    hay = gpu.synth_uniform_device(0, H, seed=1, stream=1)
    plants = [[(20 + 55 * j) * SR + 13 * j, (20 + 55 * j + 1790) * SR + 7 * j] for j in range(nn)]
    for j, n in enumerate(needles):
        for t in plants[j]:
            gpu.axpy_device(0, hay, t, n.ptr, S, 1.0)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=480.0, prominence=0.13)
    p = cfg.params(SR, gpu.Scale.LIB)
    for _ in range(2):
        res = gpu.match_multi_device(algos, hay.ptr, H, p)
        assert [[q.start for q in r] for r in res] == plants
        assert all(abs(q.height - 1.0) < 0.03 and q.prominence > 0.9 for r in res for q in r)
    for j in (0, 7, 8, 31):
        one = algos[j].match_device(hay.ptr, H, p)
        assert [(q.start, q.end) for q in one] == [(q.start, q.end) for q in res[j]]
        for a, b in zip(one, res[j]):
            assert abs(a.height - b.height) < 2e-6 and abs(a.prominence - b.prominence) < 2e-6
    for b in needles + [hay]:
        b.free()


def test_offsets_beyond_2_to_31_samples(gpu):
    """A 14 h haystack (2.22e9 samples, 8.9 GB resident): score indices, block and chunk
    arithmetic beyond 2^31, hits before and after that mark, found on the dense and the
    sparse score path."""
    hours = 14
    n = hours * H
    assert n > 2 ** 31
    needle = gpu.synth_uniform_device(0, S, seed=1, stream=0)
    algo = gpu.HipConvolve.from_device(0, needle.ptr, S)
    hay = gpu.synth_uniform_device(0, n, seed=7, stream=3)
    # more than the 480 s overshadow distance apart (audio_matcher.rs:143-160)
    plants = [12_345, 2 ** 31 - S // 2, 2 ** 31 + 600 * SR + 3, n - S - 17]
    for t in plants:
        gpu.axpy_device(0, hay, t, needle.ptr, S, 1.0)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=480.0, prominence=0.13)
    p = cfg.params(SR, gpu.Scale.LIB)
    for _ in range(2):
        got = algo.match_device(hay.ptr, n, p)
        assert [q.start for q in got] == plants
        assert all(abs(q.height - 1.0) < 0.02 and q.end == q.start + 1 for q in got)
    hay.free()
