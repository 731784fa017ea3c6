"""Randomised parity of calc_chunks (audio_matcher.rs:88-141) against the oracle over
chunk / overlap / distance / prominence combinations the fixed tests do not reach:
overlap shorter and longer than the needle, chunk lengths that are not multiples of
anything, several hits per chunk, both transform plans (generic and the N = 2^21 one)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def build_case(oracle, rng, sr, s, h, n_plants, gain_lo=0.6):
    needle = oracle.synth_uniform(int(rng.integers(1, 1 << 30)), 0, 0, s)
    hay = oracle.synth_uniform(int(rng.integers(1, 1 << 30)), 1, 0, h)
    offs = sorted(int(x) for x in rng.integers(0, max(1, h - s), size=n_plants))
    for o in offs:
        hay[o:o + s] += float(rng.uniform(gain_lo, 1.2)) * needle
    return needle, hay


def compare(gpu, oracle, needle, hay, sr, chunk, overlap, prom, dist_s, handle=None):
    p = gpu.AmMatchParams(sr=sr, chunk=chunk, overlap=overlap, min_prominence=prom,
                          min_distance=int(dist_s) * sr, overshadow_distance_s=dist_s, scale=1)
    algo = handle or gpu.HipConvolve(needle)
    exp = oracle.calc_chunks(sr, hay, needle, chunk, overlap, prom, p.min_distance, dist_s)
    for _ in range(2):   # dense scores first, sparse scores on the second call
        got = algo.match(hay, p)
        assert [(g.start, g.end) for g in got] == [(e[0], e[1]) for e in exp]
        for g, e in zip(got, exp):
            assert abs(g.height - e[2]) < TOL and abs(g.prominence - e[3]) < TOL
    return len(exp)


@pytest.mark.parametrize("seed", range(8))
def test_random_small_plans(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    sr = 8000
    s = int(rng.integers(2000, 12000))
    h = int(rng.integers(12 * s, 40 * s))
    needle, hay = build_case(oracle, rng, sr, s, h, int(rng.integers(0, 9)))
    chunk = int(rng.integers(2 * s, 9 * s)) | 1                      # odd on purpose
    overlap = int(rng.choice([0, s // 3, s - 1, s, s + 17, 2 * s]))
    prom = float(rng.choice([0.2, 0.35, 0.5]))
    dist_s = float(rng.choice([0.0, 1.0, 3.0, 30.0]))
    compare(gpu, oracle, needle, hay, sr, chunk, overlap, prom, dist_s)


@pytest.mark.parametrize("seed", range(4))
def test_random_production_plan(gpu, oracle, seed):
    """Needles of 300k-500k samples select the N = 2^21 register kernels."""
    rng = np.random.default_rng(2000 + seed)
    sr = 44100
    s = int(rng.integers(300_000, 500_000))
    h = int(rng.integers(9_000_000, 14_000_000))
    needle, hay = build_case(oracle, rng, sr, s, h, int(rng.integers(1, 6)))
    chunk = int(rng.integers(2_000_000, 4_000_000)) | 1
    overlap = int(rng.choice([s, s // 2, s + 999]))
    dist_s = float(rng.choice([5.0, 60.0, 480.0]))
    n = compare(gpu, oracle, needle, hay, sr, chunk, overlap, 0.13, dist_s)
    assert n >= 0


def test_find_peaks_random_vs_oracle(gpu, oracle):
    """am_find_peaks on random walks / noise / quantised data with random thresholds and distances."""
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = int(rng.integers(3, 120_000))
        kind = trial % 3
        if kind == 0:
            y = np.cumsum(rng.standard_normal(n)).astype(np.float32)
        elif kind == 1:
            y = rng.standard_normal(n).astype(np.float32)
        else:
            y = (np.round(rng.standard_normal(n) * 2) / 2).astype(np.float32)
        prom = float(rng.choice([0.0, 0.5, 2.0, 8.0]))
        dist = int(rng.choice([0, 1, 7, 500, 50_000]))
        exp = oracle.find_peaks(y, prom, dist, cap=max(16, n))
        got = [(p.start, p.end, p.height, p.prominence) for p in gpu.find_peaks(y, prom, dist, cap=max(16, n))]
        assert got == exp, (trial, n, prom, dist)


@pytest.mark.parametrize("seed", range(2))
def test_random_wide_plan(gpu, oracle, seed):
    """Needles of 0.6-1.2 M samples select N = 2^22 (256 x 2 x 8192 kernels), incl. hits on chunk edges."""
    rng = np.random.default_rng(3000 + seed)
    sr = 44100
    s = int(rng.integers(600_000, 1_200_000))
    h = int(rng.integers(14_000_000, 18_000_000))
    needle, hay = build_case(oracle, rng, sr, s, h, int(rng.integers(1, 4)))
    chunk = int(rng.integers(4_000_000, 6_000_000)) | 1
    # one more hit a few samples after a chunk start
    off = chunk + int(rng.integers(1, 40))
    hay[off:off + s] += needle
    compare(gpu, oracle, needle, hay, sr, chunk, s, 0.13, float(rng.choice([5.0, 480.0])))
    # level 1 on the same plan
    win = hay[: s + 3_000_000]
    algo = gpu.HipConvolve(needle)
    got = algo.correlate_with_sample(win, gpu.Mode.Valid, True)
    ref = oracle.correlate(win, needle, oracle.MODE_VALID, oracle.SCALE_LIB)
    err = np.abs(got - ref)
    assert float(err.max()) < TOL, (getattr(gpu, "gpu_identity", "?"), float(err.max()), int(np.argmax(err)), s, h)


@pytest.mark.parametrize("seed", range(8))
def test_random_chunks_shorter_than_the_needle_and_long_overlaps(gpu, oracle, seed):
    """The windowing in its less usual shapes (audio_matcher.rs:99-108 puts no constraint on them):
    chunks shorter than the needle, overlaps longer than the chunk (consecutive windows then share
    most of their scores and a hit is reported by several of them before the merge), on the generic
    plans and -- with the longer haystacks -- on the register kernels, whose chunk-edge logic takes
    its general path when a chunk is shorter than a block."""
    rng = np.random.default_rng(5000 + seed)
    sr = 8000
    s = int(rng.integers(3000, 9000))
    h = int(rng.integers(20 * s, 60 * s)) if seed % 2 == 0 else int(rng.integers(600_000, 900_000))
    needle, hay = build_case(oracle, rng, sr, s, h, int(rng.integers(2, 7)))
    chunk = int(rng.integers(s // 4, 2 * s)) | 1
    overlap = int(rng.choice([s, s + 1, 2 * s, s + chunk, s + 3 * chunk]))
    prom = float(rng.choice([0.2, 0.35]))
    dist_s = float(rng.choice([0.0, 1.0, 2.0]))
    n = compare(gpu, oracle, needle, hay, sr, chunk, overlap, prom, dist_s)
    assert n >= 1


@pytest.mark.parametrize("s", [1, 2, 3, 50, 63, 64, 65, 66, 200])
def test_needle_lengths_around_the_direct_summation_limit(gpu, oracle, s):
    """Needles of up to 64 samples are summed directly, longer ones take the transforms: every mode
    and scaling of level 1, and level 2 (windows of several chunks, pcm16 and f32 ingest), on both
    sides of the switch, against the checker."""
    rng = np.random.default_rng(7000 + s)
    sr = 8000
    needle = oracle.synth_uniform(700 + s, 0, 0, s)
    hay = oracle.synth_uniform(700 + s, 1, 0, 40 * sr)
    plants = [int(x) for x in (3.3 * sr, 17.0 * sr + 5, 29.9 * sr)]
    for o in plants:
        hay[o:o + s] += 3.0 * needle
    algo = gpu.HipConvolve(needle)
    win = hay[: 9000 + s]
    for mode, omode in ((gpu.Mode.Full, oracle.MODE_FULL), (gpu.Mode.Same, oracle.MODE_SAME), (gpu.Mode.Valid, oracle.MODE_VALID)):
        for scale, oscale in ((False, oracle.SCALE_NONE), (True, oracle.SCALE_LIB)):
            got = algo.correlate_with_sample(win, mode, scale)
            ref = oracle.correlate(win, needle, omode, oscale)
            assert got.shape == ref.shape and np.abs(got - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))
    chunk, overlap = 7 * sr + 1, s + 11
    prom = 1.5                                          # planted at gain 3: scaled score about 3
    n = compare(gpu, oracle, needle, hay, sr, chunk, overlap, prom, 2.0, handle=algo)
    assert n >= 1
    # the same through the i16 stereo ingest
    lr = np.clip(np.round(np.repeat(hay, 2) * 20000.0), -32768, 32767).astype(np.int16)
    nlr = np.clip(np.round(np.repeat(needle, 2) * 20000.0), -32768, 32767).astype(np.int16)
    a16 = gpu.HipConvolve.from_pcm16(nlr)
    m_needle, m_hay = oracle.pcm_s16_stereo_to_mono(nlr), oracle.pcm_s16_stereo_to_mono(lr)
    if float(np.sum(m_needle.astype(np.float64) ** 2)) > 0.0:
        p = gpu.AmMatchParams(sr=sr, chunk=chunk, overlap=overlap, min_prominence=prom, min_distance=2 * sr,
                              overshadow_distance_s=2.0, scale=1)
        exp = oracle.calc_chunks(sr, m_hay, m_needle, chunk, overlap, prom, p.min_distance, 2.0)
        got = a16.match_pcm16(lr, p)
        assert [(g.start, g.end) for g in got] == [(x[0], x[1]) for x in exp]
        for g, x in zip(got, exp):
            assert abs(g.height - x[2]) < TOL * 3 and abs(g.prominence - x[3]) < TOL * 3
