"""GPU parity of level 2 (calc_chunks, audio_matcher.rs:88-141) against the
oracle on seeded synthetic audio, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 44100
TOL = 1e-4


def synth_case(oracle, sr, needle_s, hay_s, plants_s, seed=1, stream=1, gain=1.0):
    s = oracle.round_samples(needle_s, sr)
    h = oracle.round_samples(hay_s, sr)
    needle = oracle.synth_uniform(seed, 0, 0, s)
    hay = oracle.synth_uniform(seed, stream, 0, h)
    for t in plants_s:
        off = oracle.round_samples(t, sr)
        n = min(s, h - off)
        hay[off:off + n] += gain * needle[:n]
    return needle, hay


def run_both(gpu, oracle, needle, hay, sr, chunk_s, overlap_s, prom, dist_s):
    cfg = gpu.Config(chunk_size_s=chunk_s, overlap_length_s=overlap_s, distance_s=dist_s, prominence=prom)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    got = algo.match(hay, p)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, prom, p.min_distance, dist_s,
                             scale=oracle.SCALE_LIB)
    return got, exp


def assert_same(got, exp):
    assert [g.start for g in got] == [e[0] for e in exp]      # identical integer offsets
    assert [g.end for g in got] == [e[1] for e in exp]
    for g, e in zip(got, exp):
        assert abs(g.height - e[2]) < TOL
        assert abs(g.prominence - e[3]) < TOL


def test_config1_10s_needle_60s_haystack(gpu, oracle):
    """BASELINE config 1: 10 s needle vs 60 s haystack, one plant at 20 s."""
    needle, hay = synth_case(oracle, SR, 10.0, 60.0, [20.0])
    got, exp = run_both(gpu, oracle, needle, hay, SR, 60.0, 10.0, 0.13, 480.0)
    assert [e[0] for e in exp] == [20 * SR]
    assert_same(got, exp)


def test_multi_chunk_with_tail_and_overshadow(gpu, oracle):
    """3.5 chunks of 20 s, plants in different chunks, two closer than `distance`
    (the weaker one is overshadowed, audio_matcher.rs:143-160)."""
    sr = 8000
    needle, hay = synth_case(oracle, sr, 2.0, 70.0, [5.0, 31.0, 64.5], seed=7)
    # weaker copy 3 s after the first plant: survives find_peaks only if in another chunk
    off = oracle.round_samples(22.0, sr)
    hay[off:off + needle.size] += 0.5 * needle
    got, exp = run_both(gpu, oracle, needle, hay, sr, 20.0, 2.0, 0.13, 25.0)
    assert len(exp) >= 2
    assert_same(got, exp)


def test_peak_on_chunk_boundary_region(gpu, oracle):
    """A plant whose peak sits in the 1-sample overlap between two chunks' score
    slices is a chunk edge in both and reported by neither; one sample later it
    is found by the second chunk."""
    sr = 8000
    for delta in (0, 1, -1):
        needle, hay = synth_case(oracle, sr, 1.0, 30.0, [], seed=3)
        off = 10 * sr + delta
        hay[off:off + needle.size] += needle
        got, exp = run_both(gpu, oracle, needle, hay, sr, 10.0, 1.0, 0.13, 480.0)
        assert_same(got, exp)


def test_no_hit_and_short_haystack(gpu, oracle):
    sr = 8000
    needle, hay = synth_case(oracle, sr, 1.0, 12.0, [], seed=5)
    got, exp = run_both(gpu, oracle, needle, hay, sr, 5.0, 1.0, 0.13, 480.0)
    assert got == [] and exp == []
    # haystack shorter than the needle: no window holds a complete needle
    got, exp = run_both(gpu, oracle, needle, hay[:needle.size - 1], sr, 5.0, 1.0, 0.13, 480.0)
    assert got == [] and exp == []


def test_small_distance_many_peaks_per_chunk(gpu, oracle):
    sr = 8000
    needle, hay = synth_case(oracle, sr, 0.5, 40.0, [1.0, 4.0, 9.5, 13.0, 22.2, 29.0, 38.0], seed=9)
    got, exp = run_both(gpu, oracle, needle, hay, sr, 15.0, 0.5, 0.4, 2.0)
    assert len(exp) == 7
    assert_same(got, exp)


def test_device_resident_and_batch(gpu, oracle):
    sr = 8000
    needle, hay1 = synth_case(oracle, sr, 1.0, 40.0, [3.0, 33.0], seed=11, stream=1)
    _, hay2 = synth_case(oracle, sr, 1.0, 25.0, [12.5], seed=11, stream=2)
    cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0, distance_s=5.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    b1 = gpu.DeviceBuffer.from_numpy(0, hay1)
    b2 = gpu.DeviceBuffer.from_numpy(0, hay2)
    res = algo.match_batch_device([b1.ptr, b2.ptr], [hay1.size, hay2.size], p)
    for r, hay in zip(res, (hay1, hay2)):
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
        assert_same(r, exp)
    assert [x.start for x in res[0]] == [3 * sr, 33 * sr]
    assert [x.start for x in res[1]] == [int(12.5 * sr)]


def test_ragged_batch_with_and_without_overlap(gpu, oracle):
    """The per-file loop over resident haystacks of very different lengths, including one
    shorter than the needle and one of exactly the needle's length: every entry equals the
    single call, with the peak pick overlapped (two buffer sets, second stream) or not."""
    sr = 44100
    needle = oracle.synth_uniform(17, 0, 0, 3 * sr)
    secs = [200.0, 1.5, 90.0, 3.0, 333.3, 47.0, 200.0, 12.0]
    hays = []
    for k, t in enumerate(secs):
        h = oracle.synth_uniform(17, 10 + k, 0, int(t * sr))
        if t > 20:
            for off in (int(0.31 * t * sr), int(0.77 * t * sr)):
                h[off:off + needle.size] += needle
        hays.append(h)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=3.0, distance_s=10.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
    singles = [algo.match_device(b.ptr, h.size, p) for b, h in zip(bufs, hays)]
    key = lambda r: [(q.start, q.end, q.height, q.prominence) for q in r]
    assert singles[1] == [] and [len(r) for r in singles] == [2, 0, 2, 0, 2, 2, 2, 0]
    try:
        for mode in (1, 0, 1):
            gpu.set_option("batch_overlap", mode)
            res = algo.match_batch_device([b.ptr for b in bufs], [h.size for h in hays], p)
            assert [key(r) for r in res] == [key(r) for r in singles], mode
    finally:
        gpu.set_option("batch_overlap", 1)


def test_synth_generator_matches_oracle_bitwise(gpu, oracle):
    n = 100_003
    buf = gpu.synth_uniform_device(0, n, seed=3, stream=17, first=12345, amp=0.25)
    got = buf.to_numpy(np.float32, n)
    exp = oracle.synth_uniform(3, 17, 12345, n, 0.25)
    assert np.array_equal(got, exp)


def test_pcm_downmix_bit_exact(gpu, oracle):
    """mp3_reader.rs:28-37: (l + r) * 0.5 * (1/65535) in f32, bit for bit."""
    rng = np.random.default_rng(0)
    lr = rng.integers(-32768, 32768, size=2 * 100_001, dtype=np.int16)
    lr[:8] = [32767, 32767, -32768, -32768, 32767, -32768, 0, 1]
    got = gpu.pcm_s16_stereo_to_mono(lr)
    exp = oracle.pcm_s16_stereo_to_mono(lr)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_sparse_score_path_after_first_call(gpu, oracle):
    """Raw scores are written only for the 32-score runs whose maximum reaches their K3 tile's
    write threshold (tile minimum, bounded by the needle's recent chunk minima, plus half a
    prominence); the threshold's history changes from call to call.  Every call must give the
    oracle's answer, also at the production transform size."""
    sr = 44100
    needle, hay = synth_case(oracle, sr, 10.0, 150.0, [20.0, 95.5, 130.0])
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=30.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    algo = gpu.HipConvolve(needle)
    first = algo.match(hay, p)
    second = algo.match(hay, p)
    third = algo.match(hay, p)
    assert len(exp) == 3
    for got in (first, second, third):
        assert_same(got, exp)


def test_theta_certificate_failure_falls_back(gpu, oracle):
    """A quiet haystack first, then a loud one whose chunk minima are far lower, then quiet again:
    the write threshold follows the tile's own minimum (no certificate failure on the way down) and
    the recent-minimum ring (on the way up); the answer equals the oracle's throughout."""
    sr = 44100
    s = 5 * sr
    needle = oracle.synth_uniform(2, 0, 0, s)
    quiet = oracle.synth_uniform(2, 1, 0, 100 * sr, 0.01)
    quiet[30 * sr:30 * sr + s] += needle
    loud = oracle.synth_uniform(2, 2, 0, 100 * sr, 2.0)
    for t in (12, 77):
        loud[t * sr:t * sr + s] += needle
    cfg = gpu.Config(chunk_size_s=40.0, overlap_length_s=5.0, distance_s=20.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    for hay in (quiet, quiet, loud, loud, quiet):
        got = algo.match(hay, p)
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 20.0)
        assert_same(got, exp)


def test_peak_in_chunk_edge_run_sparse(gpu, oracle):
    """Hits right at chunk starts / ends with the sparse path active."""
    sr = 44100
    s = 10 * sr
    needle = oracle.synth_uniform(4, 0, 0, s)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=30.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    for delta in (-33, -1, 0, 1, 2, 31, 32, 33):
        hay = oracle.synth_uniform(4, 1, 0, 200 * sr)
        off = 60 * sr + delta
        hay[off:off + s] += needle
        hay[5 * sr:5 * sr + s] += needle
        got = algo.match(hay, p)
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
        assert_same(got, exp)


def test_pcm16_stereo_ingest_fused(gpu, oracle):
    """BASELINE config 5's input format: 48 kHz interleaved i16 stereo.  The
    down-mix fused into K1 must give exactly what the separate down-mix kernel +
    f32 path gives (same f32 samples -> identical arithmetic) and the oracle's
    offsets."""
    sr = 48000
    rng = np.random.default_rng(12)
    s, h = 10 * sr, 200 * sr
    needle_lr = rng.integers(-9000, 9000, size=2 * s).astype(np.int16)
    hay_lr = rng.integers(-9000, 9000, size=2 * h).astype(np.int32)
    for t in (17.0, 71.3, 140.0):
        off = int(t * sr)
        hay_lr[2 * off:2 * (off + s)] += needle_lr
    hay_lr = np.clip(hay_lr, -32768, 32767).astype(np.int16)
    needle = oracle.pcm_s16_stereo_to_mono(needle_lr)
    hay = oracle.pcm_s16_stereo_to_mono(hay_lr)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=30.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    assert [e[0] for e in exp] == [int(17.0 * sr), int(71.3 * sr), 140 * sr]
    a_pcm = gpu.HipConvolve.from_pcm16(needle_lr)
    a_f32 = gpu.HipConvolve(needle)
    assert a_pcm.inverse_sample_auto_correlation() == a_f32.inverse_sample_auto_correlation()
    for _ in range(2):                       # second round exercises the sparse-score path
        got_pcm = a_pcm.match_pcm16(hay_lr, p)
        got_f32 = a_f32.match(hay, p)
        assert_same(got_pcm, exp)
        assert [(g.start, g.end, g.height, g.prominence) for g in got_pcm] == \
               [(g.start, g.end, g.height, g.prominence) for g in got_f32]


def test_pcm16_small_generic_plan(gpu, oracle):
    """i16 ingest through the generic kernels (small transform) incl. odd offsets."""
    sr = 8000
    rng = np.random.default_rng(3)
    s, h = 1500, 30001
    needle_lr = rng.integers(-20000, 20000, size=2 * s).astype(np.int16)
    hay_lr = rng.integers(-20000, 20000, size=2 * h).astype(np.int32)
    hay_lr[2 * 7777:2 * (7777 + s)] += needle_lr
    hay_lr = np.clip(hay_lr, -32768, 32767).astype(np.int16)
    needle = oracle.pcm_s16_stereo_to_mono(needle_lr)
    hay = oracle.pcm_s16_stereo_to_mono(hay_lr)
    cfg = gpu.Config(chunk_size_s=1.0, overlap_length_s=s / sr, distance_s=2.0, prominence=0.4)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.4, p.min_distance, 2.0)
    got = gpu.HipConvolve.from_pcm16(needle_lr).match_pcm16(hay_lr, p)
    assert [e[0] for e in exp] == [7777]
    assert_same(got, exp)


def test_multi_needle_shared_forward_pass(gpu, oracle):
    """BASELINE config 4 shape (several needles, one haystack): equals separate
    single-needle calls and the oracle, on both calls (dense and sparse scores)."""
    sr = 44100
    s, h = 5 * sr, 160 * sr
    needles = [oracle.synth_uniform(21, 100 + k, 0, s) for k in range(4)]
    hay = oracle.synth_uniform(21, 1, 0, h)
    plants = {0: [12.0, 100.0], 1: [45.5], 2: [], 3: [70.0, 120.25, 150.0]}
    for k, ts in plants.items():
        for t in ts:
            off = int(t * sr)
            hay[off:off + s] += needles[k]
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=5.0, distance_s=20.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algos = [gpu.HipConvolve(n) for n in needles]
    buf = gpu.DeviceBuffer.from_numpy(0, hay)
    exps = [oracle.calc_chunks(sr, hay, n, p.chunk, p.overlap, 0.13, p.min_distance, 20.0) for n in needles]
    for k, ts in plants.items():
        assert [e[0] for e in exps[k]] == [int(t * sr) for t in ts]
    for _ in range(2):
        res = gpu.match_multi_device(algos, buf.ptr, hay.size, p)
        for got, exp in zip(res, exps):
            assert_same(got, exp)
    singles = [a.match_device(buf.ptr, hay.size, p) for a in algos]
    # needles are processed in groups that share the forward row transforms (option
    # needle_group): every grouping, including a last group of one, gives the offsets of
    # separate calls; the group kernel rounds a few products differently (fused
    # multiply-add placement), so heights agree to f32 rounding, not bit for bit
    try:
        for group in (4, 3, 2, 1, 8):
            gpu.set_option("needle_group", group)
            res = gpu.match_multi_device(algos, buf.ptr, hay.size, p)
            for got, one in zip(res, singles):
                assert [(g.start, g.end) for g in got] == [(g.start, g.end) for g in one], group
                for g, o in zip(got, one):
                    assert abs(g.height - o.height) < 2e-6 and abs(g.prominence - o.prominence) < 2e-6, group
    finally:
        gpu.set_option("needle_group", 8)
    with pytest.raises(gpu.AudioMatchError):
        gpu.match_multi_device([algos[0], gpu.HipConvolve(needles[1][:-1])], buf.ptr, hay.size, p)


def test_handle_is_shareable_across_threads(gpu, oracle):
    """The reference shares &algo across rayon workers (C: Sync, audio_matcher.rs:89,114-122):
    concurrent calls on one handle, and on two handles, must stay correct."""
    import threading
    sr = 8000
    needle, hay_a = synth_case(oracle, sr, 1.0, 40.0, [3.0, 33.0], seed=31, stream=1)
    _, hay_b = synth_case(oracle, sr, 1.0, 25.0, [12.5], seed=31, stream=2)
    cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0, distance_s=5.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    other = gpu.HipConvolve(needle[::-1].copy())
    exp_a = oracle.calc_chunks(sr, hay_a, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
    exp_b = oracle.calc_chunks(sr, hay_b, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
    exp_win = oracle.correlate(hay_a[:20000], needle, oracle.MODE_VALID, oracle.SCALE_LIB)
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                if i % 3 == 0:
                    assert_same(algo.match(hay_a, p), exp_a)
                elif i % 3 == 1:
                    assert_same(algo.match(hay_b, p), exp_b)
                else:
                    got = algo.correlate_with_sample(hay_a[:20000], gpu.Mode.Valid, True)
                    assert np.abs(got - exp_win).max() < TOL
                    assert other.match(hay_b, p) == []
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_error_codes(gpu):
    import ctypes as C
    L = gpu.lib()
    algo = gpu.HipConvolve(np.ones(8, np.float32))
    n = C.c_size_t(0)
    out = (C.c_float * 4)()
    x = np.ones(100, np.float32)
    # capacity too small: required length is reported
    rc = L.am_correlate(algo._h, x.ctypes.data, x.size, 2, 0, out, 4, C.byref(n))
    assert rc == gpu.AM_ERR_CAPACITY and n.value == 93
    assert L.am_correlate(algo._h, x.ctypes.data, x.size, 7, 0, out, 4, C.byref(n)) == gpu.AM_ERR_INVALID_ARG
    assert L.am_correlate(None, x.ctypes.data, x.size, 2, 0, out, 4, C.byref(n)) == gpu.AM_ERR_INVALID_ARG
    p = gpu.Config().params(8000, gpu.Scale.MY)     # any CorrelateAlgo's scaling rides calc_chunks (audio_matcher.rs:88-97)
    buf = (gpu.AmPeak * 4)()
    assert L.am_match(algo._h, x.ctypes.data, x.size, C.byref(p), buf, 4, C.byref(n)) == 0
    p.scale = 3
    assert L.am_match(algo._h, x.ctypes.data, x.size, C.byref(p), buf, 4, C.byref(n)) == gpu.AM_ERR_INVALID_ARG
    p = gpu.Config().params(8000, gpu.Scale.LIB)
    p.chunk = 0
    assert L.am_match(algo._h, x.ctypes.data, x.size, C.byref(p), buf, 4, C.byref(n)) == gpu.AM_ERR_INVALID_ARG
    h = C.c_void_p()
    assert L.am_needle_create(99, x.ctypes.data, 8, C.byref(h)) == gpu.AM_ERR_NO_DEVICE
    assert L.am_needle_create(0, x.ctypes.data, 0, C.byref(h)) == gpu.AM_ERR_INVALID_ARG
    # peak-capacity overflow on am_match: count is reported
    sr = 8000
    y = np.zeros(4 * sr, np.float32)
    spike = gpu.HipConvolve(np.array([1.0], np.float32))
    y[[1000, 9000, 17000, 25000]] = 1.0
    cfg = gpu.Config(chunk_size_s=1.0, overlap_length_s=0.0, distance_s=0.0, prominence=0.5)
    pp = cfg.params(sr, gpu.Scale.LIB)
    small = (gpu.AmPeak * 2)()
    rc = L.am_match(spike._h, y.ctypes.data, y.size, C.byref(pp), small, 2, C.byref(n))
    assert rc == gpu.AM_ERR_CAPACITY and n.value == 4


@pytest.mark.timeout(120)
def test_degenerate_signals_terminate(gpu, oracle):
    """Inputs the reference would choke on must not hang or crash the device: a silent
    haystack (every score exactly 0: one plateau, no peak), a silent needle (energy 0:
    every scaled score is NaN) and a haystack with NaNs / infinities in one place (the windows
    that hold them lose their peaks, as in the reference; every other hit is still found --
    tests/test_gpu_round2.py::test_non_finite_samples_cost_only_their_own_windows checks that
    against the checker window by window)."""
    sr = 44100
    s = 2 * sr
    needle = oracle.synth_uniform(5, 0, 0, s)
    cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=2.0, distance_s=5.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    # both plans: the register kernels (span > 2^19) and the generic ones
    for h in (400 * sr, 9 * sr):
        silent = np.zeros(h, np.float32)
        for _ in range(2):                       # dense, then sparse score path
            assert algo.match(silent, p) == []
        hay = oracle.synth_uniform(5, 1, 0, h)
        off = h - 4 * sr
        hay[off:off + s] += needle
        bad = hay.copy()
        bad[1000] = np.nan
        bad[1001] = np.inf
        bad[1002] = -np.inf
        for _ in range(2):
            got = algo.match(bad, p)
            assert isinstance(got, list)
            if h > 40 * sr:                      # the poisoned blocks end long before the hit
                assert off in [g.start for g in got]
    zero = gpu.HipConvolve(np.zeros(s, np.float32))
    for _ in range(2):
        assert isinstance(zero.match(oracle.synth_uniform(5, 2, 0, 30 * sr), p), list)


@pytest.mark.parametrize("level", [1, 2])
def test_half_pipeline_config5(gpu, oracle, level):
    """BASELINE config 5: 48 kHz interleaved i16 stereo through the half-precision
    pipeline (level 1: work matrix stored as f16, f32 butterflies; level 2: K2's butterflies
    in packed f16 as well).  SURVEY 7: offset parity is the promise; scores are additionally
    checked to 1e-3."""
    sr = 48000
    rng = np.random.default_rng(77)
    s, h = 10 * sr, 250 * sr
    needle_lr = rng.integers(-9000, 9000, size=2 * s).astype(np.int16)
    hay_lr = rng.integers(-9000, 9000, size=2 * h).astype(np.int32)
    for t in (17.0, 71.3, 140.0, 222.2):
        off = int(t * sr)
        hay_lr[2 * off:2 * (off + s)] += needle_lr
    hay_lr = np.clip(hay_lr, -32768, 32767).astype(np.int16)
    needle = oracle.pcm_s16_stereo_to_mono(needle_lr)
    hay = oracle.pcm_s16_stereo_to_mono(hay_lr)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=30.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    assert len(exp) == 4
    gpu.set_option("half_pipeline", level)
    try:
        algo = gpu.HipConvolve.from_pcm16(needle_lr)
        for _ in range(2):
            got = algo.match_pcm16(hay_lr, p)
            assert [g.start for g in got] == [e[0] for e in exp]          # identical integer offsets
            for g, e in zip(got, exp):
                assert abs(g.height - e[2]) < 1e-3 and abs(g.prominence - e[3]) < 1e-3
        # level 1 through the same pipeline: bounded error on the whole score vector
        win = hay[: 3 * 1024 * 1024]
        sc = algo.correlate_with_sample(win, gpu.Mode.Valid, True)
        ref = oracle.correlate(win, needle, oracle.MODE_VALID, oracle.SCALE_LIB)
        assert np.abs(sc - ref).max() < 1e-3
        err = float(np.abs(sc - ref).max())
    finally:
        gpu.set_option("half_pipeline", 0)
    print("half pipeline level", level, "max score error", err)


def test_shutdown_releases_and_recovers(gpu, oracle):
    sr = 8000
    needle, hay = synth_case(oracle, sr, 1.0, 30.0, [4.0, 17.0], seed=41)
    cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0, distance_s=5.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    before = algo.match(hay, p)
    assert gpu.lib().am_shutdown() == 0
    after = algo.match(hay, p)          # plans and scratch are rebuilt on demand
    assert [(q.start, q.height) for q in before] == [(q.start, q.height) for q in after]
    assert [q.start for q in after] == [4 * sr, 17 * sr]
