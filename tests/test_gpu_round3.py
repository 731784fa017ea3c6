"""GPU tests of the round-3 boundary work, all through the C ABI:
several needles x a batch of haystacks (am_match_multi_batch_device, BASELINE configs[3]), the
multi-needle and the i16-stereo pools (am_pool_create_multi, am_pool_match_multi_batch*,
am_pool_match_batch_pcm16*), and the non-finite-sample corner cases of the round-2 review."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def key(r):
    return [(q.start, q.end, q.height, q.prominence) for q in r]


def pos(r):
    return [(q.start, q.end) for q in r]


def assert_same(got, exp, tol=TOL):
    assert [g.start for g in got] == [e[0] for e in exp]
    assert [g.end for g in got] == [e[1] for e in exp]
    for g, e in zip(got, exp):
        assert abs(g.height - e[2]) < tol and abs(g.prominence - e[3]) < tol


def assert_close_peaks(got, one, tol=2e-6):
    """Same peaks as a separate single-needle call: positions identical, values to f32 rounding
    (the needle-group row kernel places a few fused multiply-adds differently)."""
    assert pos(got) == pos(one)
    for g, o in zip(got, one):
        assert abs(g.height - o.height) < tol and abs(g.prominence - o.prominence) < tol


# ---------------------------------------------------------------------------
# several needles x several haystacks
# ---------------------------------------------------------------------------
def multi_inputs(oracle, sr, n_needles=3):
    """Ragged batch: a long haystack (register kernels), a short one (generic plan), one shorter
    than the needle, an empty one, another long one; needle j planted at its own offsets."""
    s = 3 * sr
    needles = [oracle.synth_uniform(31, 200 + j, 0, s) for j in range(n_needles)]
    lens = [170 * sr, 25 * sr, s - 5, 0, 95 * sr + 321]
    plants = {0: {0: [12.0, 100.5], 1: [61.0], 2: []},
              1: {0: [], 1: [20.25], 2: [5.0]},
              4: {0: [33.0], 1: [], 2: [70.0, 88.0]}}
    hays = []
    for k, n in enumerate(lens):
        h = oracle.synth_uniform(31, 1 + k, 0, n) if n else np.zeros(0, np.float32)
        for j, ts in plants.get(k, {}).items():
            if j < n_needles:
                for t in ts:
                    off = int(t * sr)
                    h[off:off + s] += needles[j]
        hays.append(h)
    return needles, hays, plants


def test_multi_batch_equals_single_calls_and_oracle(gpu, oracle):
    """am_match_multi_batch_device (matcher::run's file loop around N snippets, matcher/mod.rs:42-87):
    every (haystack, needle) pair equals the separate single-needle call -- positions identical,
    values to f32 rounding -- and the checker; with and without the overlapped peak pick, for
    every needle grouping, twice (dense first call, sparse afterwards)."""
    sr = 16000
    needles, hays, plants = multi_inputs(oracle, sr)
    cfg = gpu.Config(chunk_size_s=30.0, overlap_length_s=3.0, distance_s=10.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algos = [gpu.HipConvolve(n) for n in needles]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) if h.size else None for h in hays]
    ptrs = [b.ptr if b else None for b in bufs]
    lens = [h.size for h in hays]
    singles = [[a.match_device(b.ptr, h.size, p) if b is not None else [] for a in algos] for b, h in zip(bufs, hays)]
    for k, per in plants.items():
        for j, ts in per.items():
            assert [q.start for q in singles[k][j]] == [int(t * sr) for t in ts], (k, j)
    exp = oracle.calc_chunks(sr, hays[0], needles[0], p.chunk, p.overlap, 0.13, p.min_distance, 10.0)
    try:
        for overlap in (1, 0):
            gpu.set_option("batch_overlap", overlap)
            for group in (8, 2, 1):
                gpu.set_option("needle_group", group)
                for _ in range(2):
                    res = gpu.match_multi_batch_device(algos, ptrs, lens, p)
                    assert len(res) == len(hays) and all(len(r) == len(algos) for r in res)
                    for k in range(len(hays)):
                        for j in range(len(algos)):
                            assert_close_peaks(res[k][j], singles[k][j])
                    assert_same(res[0][0], exp)
    finally:
        gpu.set_option("batch_overlap", 1)
        gpu.set_option("needle_group", 8)
    # one haystack through the batch entry point == am_match_multi_device
    one = gpu.match_multi_device(algos, ptrs[0], lens[0], p)
    assert [key(r) for r in one] == [key(r) for r in gpu.match_multi_batch_device(algos, ptrs[:1], lens[:1], p)[0]]
    # capacity: counts are reported, AM_ERR_CAPACITY returned
    nn, k = len(algos), len(hays)
    handles = (C.c_void_p * nn)(*[a._h for a in algos])
    counts = (C.c_size_t * (nn * k))()
    rc = gpu.lib().am_match_multi_batch_device(handles, nn, (C.c_void_p * k)(*ptrs), (C.c_size_t * k)(*lens), k, 0, C.byref(p),
                                               None, 0, counts)
    assert rc == gpu.AM_ERR_CAPACITY
    assert [counts[i * nn + j] for i in range(k) for j in range(nn)] == [len(singles[i][j]) for i in range(k) for j in range(nn)]
    rc = gpu.lib().am_match_multi_batch_device(handles, nn, (C.c_void_p * k)(*ptrs), (C.c_size_t * k)(*lens), k, 7, C.byref(p),
                                               None, 0, counts)
    assert rc == gpu.AM_ERR_INVALID_ARG


def test_multi_batch_pcm16_and_half_levels(gpu, oracle):
    """The same loop on interleaved i16 stereo frames (mp3_reader.rs:26-37) -- BASELINE configs[3]
    and [4] combined -- at every precision level: offsets of the separate calls."""
    sr = 16000
    rng = np.random.default_rng(3)
    s = 2 * sr
    needles_lr = [rng.integers(-9000, 9000, size=2 * s).astype(np.int16) for _ in range(2)]
    hays_lr = []
    plant = {0: {0: [11.0, 70.0], 1: [40.5]}, 1: {0: [], 1: [15.0]}}
    for k, secs in enumerate((100, 36)):
        h = rng.integers(-9000, 9000, size=2 * secs * sr).astype(np.int32)
        for j, ts in plant[k].items():
            for t in ts:
                off = int(t * sr)
                h[2 * off:2 * (off + s)] += needles_lr[j]
        hays_lr.append(np.clip(h, -32768, 32767).astype(np.int16))
    p = gpu.Config(chunk_size_s=20.0, overlap_length_s=2.0, distance_s=8.0, prominence=0.4).params(sr, gpu.Scale.LIB)
    algos = [gpu.HipConvolve.from_pcm16(n) for n in needles_lr]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays_lr]
    frames = [h.size // 2 for h in hays_lr]
    singles = [[a.match_pcm16_device(b.ptr, f, p) for a in algos] for b, f in zip(bufs, frames)]
    for k, per in plant.items():
        for j, ts in per.items():
            assert [q.start for q in singles[k][j]] == [int(t * sr) for t in ts]
    res = gpu.match_multi_batch_device(algos, [b.ptr for b in bufs], frames, p, fmt=gpu.Fmt.S16_STEREO)
    for k in range(2):
        for j in range(2):
            assert_close_peaks(res[k][j], singles[k][j])
    try:
        for level in (1, 2):
            gpu.set_option("half_pipeline", level)
            res = gpu.match_multi_batch_device(algos, [b.ptr for b in bufs], frames, p, fmt=gpu.Fmt.S16_STEREO)
            for k in range(2):
                for j in range(2):
                    assert pos(res[k][j]) == pos(singles[k][j])
                    for g, o in zip(res[k][j], singles[k][j]):
                        assert abs(g.height - o.height) < 2e-3
    finally:
        gpu.set_option("half_pipeline", 0)


def test_multi_batch_non_finite_samples(gpu, oracle):
    """A NaN in one haystack of a several-needle batch: exactly the windows that hold it lose their
    peaks, for every needle, as in the reference (audio_matcher.rs:114-122) and in am_match_device;
    the other haystacks of the batch are untouched and nothing sticks to the handles."""
    sr = 8000
    s = 2 * sr
    needles = [oracle.synth_uniform(41, 10 + j, 0, s) for j in range(2)]
    hay = oracle.synth_uniform(41, 1, 0, 100 * sr)
    for j, ts in enumerate(((13, 35, 57, 81), (5, 33, 91))):
        for t in ts:
            hay[t * sr:t * sr + s] += needles[j]
    bad = hay.copy()
    bad[34 * sr] = np.nan          # window 3 (30 .. 42 s) of 10 s chunks with 2 s of overlap
    bad[34 * sr + 9] = -np.inf
    p = gpu.Config(chunk_size_s=10.0, overlap_length_s=2.0, distance_s=5.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algos = [gpu.HipConvolve(n) for n in needles]
    exp_clean = [oracle.calc_chunks(sr, hay, n, p.chunk, p.overlap, 0.13, p.min_distance, 5.0) for n in needles]
    exp_bad = [oracle.calc_chunks(sr, bad, n, p.chunk, p.overlap, 0.13, p.min_distance, 5.0) for n in needles]
    assert [len(e) for e in exp_clean] == [4, 3] and [len(e) for e in exp_bad] == [3, 2]
    bufs = [gpu.DeviceBuffer.from_numpy(0, x) for x in (hay, bad, hay)]
    for _ in range(2):
        res = gpu.match_multi_batch_device(algos, [b.ptr for b in bufs], [hay.size] * 3, p)
        for k, exps in enumerate((exp_clean, exp_bad, exp_clean)):
            for j in range(2):
                assert_same(res[k][j], exps[j])
    # the single-haystack form as well (match_multi had no non-finite handling)
    res = gpu.match_multi_device(algos, bufs[1].ptr, hay.size, p)
    for j in range(2):
        assert_same(res[j], exp_bad[j])
        assert_same(algos[j].match_device(bufs[0].ptr, hay.size, p), exp_clean[j])


# ---------------------------------------------------------------------------
# pools
# ---------------------------------------------------------------------------
def test_multi_pool_equals_multi_batch(gpu, oracle):
    """am_pool_create_multi / am_pool_match_multi_batch*: one, two and three slots on device 0 give
    the results of am_match_multi_batch_device bit for bit, from host and from resident buffers;
    the single-needle pool calls refuse a pool of several needles."""
    sr = 16000
    needles, hays, _ = multi_inputs(oracle, sr)
    p = gpu.Config(chunk_size_s=30.0, overlap_length_s=3.0, distance_s=10.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algos = [gpu.HipConvolve(n) for n in needles]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) if h.size else None for h in hays]
    ptrs = [b.ptr if b else None for b in bufs]
    lens = [h.size for h in hays]
    gpu.match_multi_batch_device(algos, ptrs, lens, p)
    ref = gpu.match_multi_batch_device(algos, ptrs, lens, p)
    assert sum(len(r) for per in ref for r in per) == 8
    for devices in ([0], [0, 0], [0, 0, 0]):
        pool = gpu.MultiPool(needles, devices)
        assert pool.size == len(devices) and pool.n_needles == len(needles)
        pool.match_batch_device(ptrs, lens, p)
        for _ in range(2):
            got = pool.match_batch_device(ptrs, lens, p)
            assert [[key(r) for r in per] for per in got] == [[key(r) for r in per] for per in ref], devices
            got = pool.match_batch(hays, p)                   # host buffers: ring + copier thread per slot
            assert [[key(r) for r in per] for per in got] == [[key(r) for r in per] for per in ref], devices
        assert pool.match_batch([], p) == []
        k = len(hays)
        counts = (C.c_size_t * k)()
        rc = gpu.lib().am_pool_match_batch_device(pool._p, (C.c_void_p * k)(*ptrs), (C.c_size_t * k)(*lens), k, C.byref(p),
                                                  None, 0, counts)
        assert rc == gpu.AM_ERR_INVALID_ARG and b"several needles" in gpu.lib().am_last_error_string()
        pool.close()
    n = C.c_size_t(0)
    one = gpu.Pool(needles[0], [0])
    assert gpu.lib().am_pool_needle_count(one._p, C.byref(n)) == 0 and n.value == 1


def test_pool_pcm16_equals_single_calls(gpu, oracle):
    """am_pool_match_batch_pcm16 / _device: the file loop on the format the reference decodes to
    (mp3_reader.rs:26-37), haystack k on slot k mod n: equal to am_match_pcm16 per haystack."""
    sr = 16000
    rng = np.random.default_rng(9)
    s = 2 * sr
    needle_lr = rng.integers(-9000, 9000, size=2 * s).astype(np.int16)
    hays = []
    for secs, ts in ((90, (10.0, 61.5)), (0, ()), (35, (21.0,)), (1, ()), (120, (5.0, 50.0, 110.0))):
        h = rng.integers(-9000, 9000, size=2 * secs * sr).astype(np.int32)
        for t in ts:
            off = int(t * sr)
            h[2 * off:2 * (off + s)] += needle_lr
        hays.append(np.clip(h, -32768, 32767).astype(np.int16))
    p = gpu.Config(chunk_size_s=20.0, overlap_length_s=2.0, distance_s=8.0, prominence=0.4).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve.from_pcm16(needle_lr)
    singles = [key(algo.match_pcm16(h, p)) if h.size else [] for h in hays]
    assert [len(r) for r in singles] == [2, 0, 1, 0, 3]
    mono = gpu.pcm_s16_stereo_to_mono(needle_lr)
    for devices in ([0], [0, 0], [0, 0, 0]):
        pool = gpu.Pool(mono, devices)
        for _ in range(2):
            assert [key(r) for r in pool.match_batch_pcm16(hays, p)] == singles, devices
        bufs = [gpu.DeviceBuffer.from_numpy(0, h) if h.size else None for h in hays]
        res = pool.match_batch_pcm16_device([b.ptr if b else None for b in bufs], [h.size // 2 for h in hays], p)
        assert [key(r) for r in res] == singles, devices
        pool.close()
    # half_pipeline 2 (BASELINE configs[4]) through the pool: a process-wide default the pool's handles follow
    try:
        gpu.set_option("half_pipeline", 2)
        pool = gpu.Pool(mono, [0, 0])
        res = pool.match_batch_pcm16(hays, p)
        assert [[(q[0], q[1]) for q in key(r)] for r in res] == [[(q[0], q[1]) for q in r] for r in singles]
        pool.close()
    finally:
        gpu.set_option("half_pipeline", 0)


# ---------------------------------------------------------------------------
# non-finite samples: the two corner cases of the round-2 review
# ---------------------------------------------------------------------------
def slack_case(oracle, sr, secs, hop, n_fft, s, chunk, seed):
    """A NaN in the samples a block pair READS beyond the ones its scores depend on (hop is rounded
    down to a multiple of 1024, a block still loads N samples), and a plant whose lag is the
    second-to-last score of a clean window that has to be correlated again."""
    needle = oracle.synth_uniform(seed, 0, 0, s)
    hay = oracle.synth_uniform(seed, 1, 0, secs * sr)
    # pair 0 = blocks 0, 1 reads samples [0, hop + N); its scores [0, 2 hop) depend on [0, 2 hop + s - 1)
    lo, hi = 2 * hop + s - 1, hop + n_fft
    assert hi - lo > 100
    nan_at = lo + 100
    w_bad = [i for i in range(secs * sr // chunk + 1) if i * chunk <= nan_at < i * chunk + chunk + s]
    clean_fed_by_pair0 = [i for i in range(w_bad[0]) if i * chunk < 2 * hop]
    assert len(clean_fed_by_pair0) >= 2
    plants = [clean_fed_by_pair0[0] * chunk + 3 * sr,
              clean_fed_by_pair0[-1] * chunk + chunk - 1,            # second-to-last score of that window
              (w_bad[-1] + 2) * chunk + 4 * sr]
    for t in plants:
        hay[t:t + s] += needle
    bad = hay.copy()
    bad[nan_at] = np.nan
    return needle, hay, bad, plants


@pytest.mark.parametrize("plan", ["generic_2^17", "register_2^21"])
def test_nan_in_the_rounding_slack_and_peak_at_window_end(gpu, oracle, plan):
    """(a) classify_nonfinite must use the sample range K1 actually reads, (b) a
    window that is correlated again keeps its LAST score (a peak at the second-to-last one stays a
    peak).  Expected result: the checker's, window by window (audio_matcher.rs:114-124)."""
    sr = 8000
    s = 2 * sr
    chunk = 10 * sr
    if plan == "generic_2^17":
        log_n, secs = 17, 100
    else:
        log_n, secs = 21, 560
    n_fft = 1 << log_n
    hop = (n_fft - s + 1) // 1024 * 1024
    needle, hay, bad, plants = slack_case(oracle, sr, secs, hop, n_fft, s, chunk, 51)
    p = gpu.Config(chunk_size_s=10.0, overlap_length_s=2.0, distance_s=5.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    assert p.chunk == chunk and p.overlap == s
    algo = gpu.HipConvolve(needle)
    algo.set_option("log_n", log_n)
    exp_clean = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
    exp_bad = oracle.calc_chunks(sr, bad, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
    assert [e[0] for e in exp_clean] == plants and [e[0] for e in exp_bad] == plants
    for _ in range(2):
        assert_same(algo.match(bad, p), exp_bad)
    assert_same(algo.match(hay, p), exp_clean)


# ---------------------------------------------------------------------------
# streaming ingest
# ---------------------------------------------------------------------------
def push_ragged(stream, data, sizes, per=1):
    """Push `data` in pieces of the given sizes (in elements; the last size repeats)."""
    off, i = 0, 0
    n = data.size // per
    while off < n:
        k = min(sizes[min(i, len(sizes) - 1)], n - off)
        stream.push(data[per * off:per * (off + k)])
        off += k
        i += 1


def test_stream_ingest_equals_am_match(gpu, oracle):
    """am_match_stream_begin / push / finish (calc_chunks on the reference's lazy sample iterator,
    audio_matcher.rs:88-104, mp3_reader.rs:13-41): ragged pushes give am_match's result bit for bit --
    with the length announced (block pairs are transformed while later samples arrive), announced
    too short (the buffer grows and the early pairs are computed again), unknown, on a haystack too
    short for any early pair, on a forced generic plan; the stream object is reused for a second file."""
    sr = 8000
    s = 2 * sr
    needle = oracle.synth_uniform(61, 0, 0, s)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=2.0, distance_s=30.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)

    def make(secs, plants, seed):
        hay = oracle.synth_uniform(seed, 1, 0, secs * sr)
        for t in plants:
            off = int(t * sr)
            hay[off:off + s] += needle
        return hay

    long_hay = make(1150, (17.0, 300.5, 519.9, 520.1, 1000.0, 1147.5), 62)      # 9.2 M samples: five blocks of the 2^21 plan
    other = make(1100, (1.0, 777.0), 63)
    short = make(100, (13.0, 81.0), 64)
    want = {id(h): key(algo.match(h, p)) for h in (long_hay, other, short)}
    exp = oracle.calc_chunks(sr, long_hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    assert [e[0] for e in exp] == [q[0] for q in want[id(long_hay)]]
    for expected_len in (long_hay.size, 5000000, 0):
        st = gpu.MatchStream(algo, p, expected_len)
        push_ragged(st, long_hay, [1, 999, 1234567, 3000001, 77, 2500000])
        assert key(st.finish()) == want[id(long_hay)], expected_len
        push_ragged(st, other, [4000000, 123])                  # reuse: the next file
        assert key(st.finish()) == want[id(other)], expected_len
        push_ragged(st, short, [100000])
        assert key(st.finish()) == want[id(short)], expected_len
        assert st.finish() == []                                 # nothing pushed
        st.close()
    # early pairs really ran in the announced case: the K1 launches of a stream equal those of am_match
    gpu.set_option("profile_mask", -1)
    with gpu.Profile(0) as prof:
        st = gpu.MatchStream(algo, p, long_hay.size)
        push_ragged(st, long_hay, [3000000])
        after_push = prof.query("k1_cols_fwd")[1]
        res = st.finish()
        total = prof.query("k1_cols_fwd")[1]
        st.close()
    assert key(res) == want[id(long_hay)] and after_push >= 1 and total > after_push
    # a NaN that arrives in the middle of a stream whose early pairs are already transformed
    bad = long_hay.copy()
    bad[4160000] = np.nan                                   # 520 s: inside the window that holds two of the plants
    st = gpu.MatchStream(algo, p, bad.size)
    push_ragged(st, bad, [3000000])
    got_bad = key(st.finish())
    st.close()
    assert got_bad == key(algo.match(bad, p)) and len(got_bad) < len(want[id(long_hay)])
    # a forced generic plan (no early pairs) and MyConvolve scaling go through finish alone
    forced = gpu.HipConvolve(needle)
    forced.set_option("log_n", 17)
    st = gpu.MatchStream(forced, p, short.size)
    push_ragged(st, short, [33333])
    assert key(st.finish()) == key(forced.match(short, p))
    st.close()
    pm = gpu.Config(chunk_size_s=60.0, overlap_length_s=2.0, distance_s=30.0, prominence=1e-7).params(sr, gpu.Scale.MY)
    st = gpu.MatchStream(algo, pm, short.size)
    push_ragged(st, short, [50000])
    assert key(st.finish()) == key(algo.match(short, pm))
    st.close()


def test_stream_ingest_pcm16(gpu, oracle):
    """The same on interleaved i16 stereo frames, the format the decoder yields (mp3_reader.rs:26-37)."""
    sr = 16000
    rng = np.random.default_rng(17)
    s = 2 * sr
    needle_lr = rng.integers(-9000, 9000, size=2 * s).astype(np.int16)
    h = rng.integers(-9000, 9000, size=2 * 560 * sr).astype(np.int32)       # 9 M frames
    for t in (10.0, 333.3, 555.0):
        off = int(t * sr)
        h[2 * off:2 * (off + s)] += needle_lr
    hay = np.clip(h, -32768, 32767).astype(np.int16)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=2.0, distance_s=30.0, prominence=0.4).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve.from_pcm16(needle_lr)
    want = key(algo.match_pcm16(hay, p))
    assert [q[0] for q in want] == [int(t * sr) for t in (10.0, 333.3, 555.0)]
    st = gpu.MatchStream(algo, p, hay.size // 2, fmt=gpu.Fmt.S16_STEREO)
    push_ragged(st, hay, [1152, 1152 * 1000, 2000001], per=2)       # MP3 frames are 1152 samples
    assert key(st.finish()) == want
    st.close()
    with pytest.raises(gpu.AudioMatchError):
        gpu.MatchStream(algo, p, 10, fmt=5)


def test_short_haystack_takes_the_small_plan(gpu, oracle):
    """BASELINE configs[0] (10 s needle, one 60 s window): the scores fit one pair of 2^21 blocks, so the
    library does not run a pair of 2^22 (half the points); the result equals the checker's and the
    forced 2^22 plan's offsets."""
    sr = 44100
    needle = oracle.synth_uniform(71, 0, 0, 10 * sr)
    hay = oracle.synth_uniform(71, 1, 0, 60 * sr)
    hay[20 * sr:30 * sr] += needle
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=480.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 480.0)
    algo = gpu.HipConvolve(needle)
    got = algo.match(hay, p)
    assert_same(got, exp)
    wide = gpu.HipConvolve(needle)
    wide.set_option("log_n", 22)
    assert pos(wide.match(hay, p)) == pos(got) == [(20 * sr, 20 * sr + 1)]


# ---------------------------------------------------------------------------
# long needles: needle partitioning
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("s", [3300000, 4500000])
def test_long_needles(gpu, oracle, s):
    """MyConvolve::correlate takes a needle of any length (audio_matcher.rs:414-457).  Needles above
    1.6 M samples run on N = 2^23 (3.3 M here); above 2^22 samples (4.5 M here) the needle is cut into
    segments of at most 2^22 samples whose correlations -- each on a source shifted by the segment's
    offset -- K3 adds up in the score array.  Level 1 (all three modes), calc_chunks, a batch, the
    several-needle entry point, a stream and a NaN in the haystack against the checker."""
    sr = 8000
    needle = oracle.synth_uniform(81, 0, 0, s)
    hay = oracle.synth_uniform(81, 1, 0, 14000000)
    plants = [400000, 8000000]
    for t in plants:
        hay[t:t + s] += needle
    algo = gpu.HipConvolve(needle)
    inv = 1.0 / float(np.sum(needle.astype(np.float64) ** 2))
    assert abs(algo.inverse_sample_auto_correlation() - inv) < 1e-6 * inv
    # level 1
    within = hay[:6000000]
    for mode, omode in ((gpu.Mode.Valid, oracle.MODE_VALID), (gpu.Mode.Same, oracle.MODE_SAME), (gpu.Mode.Full, oracle.MODE_FULL)):
        got = algo.correlate_with_sample(within, mode, True)
        exp = oracle.correlate(within, needle, omode, oracle.SCALE_LIB)
        assert got.shape == exp.shape and float(np.abs(got - exp).max()) < TOL
    # level 2: chunks of 600 s with the needle's length as overlap (Config::from_args, audio_matcher.rs:41)
    p = gpu.AmMatchParams(sr=sr, chunk=600 * sr, overlap=s, min_prominence=0.13, min_distance=480 * sr,
                          overshadow_distance_s=480.0, scale=int(gpu.Scale.LIB))
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 480.0)
    assert [e[0] for e in exp] == plants
    for _ in range(2):
        assert_same(algo.match(hay, p), exp)
    buf = gpu.DeviceBuffer.from_numpy(0, hay)
    res = algo.match_batch_device([buf.ptr, buf.ptr], [hay.size] * 2, p)
    assert_same(res[0], exp)
    assert key(res[0]) == key(res[1])
    # the several-needle entry point takes such needles too
    res = gpu.match_multi_batch_device([algo, algo], [buf.ptr], [hay.size], p)
    assert pos(res[0][0]) == pos(res[0][1]) == [(t, t + 1) for t in plants]
    # a stream (no early pairs for such a needle) and a NaN in the haystack
    st = gpu.MatchStream(algo, p, hay.size)
    push_ragged(st, hay, [2500000])
    assert_same(st.finish(), exp)
    st.close()
    bad = hay.copy()
    bad[13000000] = np.nan                                  # only the last two windows (from 9.6 M on) hold it
    exp_bad = oracle.calc_chunks(sr, bad, needle, p.chunk, p.overlap, 0.13, p.min_distance, 480.0)
    assert_same(algo.match(bad, p), exp_bad)


def test_needle_above_the_largest_transform(gpu, oracle):
    """8.5 M samples -- more than 2^23, which no single transform of the library could hold: three segments."""
    s2 = 8500000
    needle2 = oracle.synth_uniform(82, 0, 0, s2)
    within2 = oracle.synth_uniform(82, 1, 0, 12000000)
    within2[1234567:1234567 + s2] += needle2
    got = gpu.HipConvolve(needle2).correlate_with_sample(within2, gpu.Mode.Valid, True)
    exp = oracle.correlate(within2, needle2, oracle.MODE_VALID, oracle.SCALE_LIB)
    assert got.shape == exp.shape == (12000000 - s2 + 1,)
    assert float(np.abs(got - exp).max()) < TOL and int(np.argmax(got)) == 1234567
    forced = gpu.HipConvolve(needle2)
    forced.set_option("log_n", 23)
    with pytest.raises(gpu.AudioMatchError):
        forced.correlate_with_sample(within2, gpu.Mode.Valid, True)


# ---------------------------------------------------------------------------
# the non-white signals of bench.py against the checker
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["non_white_signal", "non_white_ar1", "non_white_speechlike", "hits_with_negative_lobes"])
def test_non_white_generators_against_the_checker(gpu, oracle, name):
    """bench.py's three non-white signals (tone and drift, AR(1) noise, AR(1) under a speech-like
    envelope) at 20 minutes of 44.1 kHz audio with the 10 s needle: offsets, plateau ends, heights and
    prominences of am_match_device equal the checker's (audio_matcher.rs:88-141), as a single call and
    inside a batch."""
    import bench
    sr = 44100
    s, h = 10 * sr, 1200 * sr
    nbuf, algo, hbuf, plants, _ = bench.NON_WHITE[name](gpu, 0, s, h)
    assert len(plants) == 2
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=480.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    needle, hay = nbuf.to_numpy("float32", s), hbuf.to_numpy("float32", h)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 480.0)
    assert [e[0] for e in exp] == plants
    got = algo.match_device(hbuf.ptr, h, p)
    assert_same(got, exp)
    res = algo.match_batch_device([hbuf.ptr] * 3, [h] * 3, p)
    assert [key(r) for r in res] == [key(got)] * 3
    # A smaller prominence bound and no distance filter: up to half a million peaks (the ripple crests of
    # the tone-and-drift signal).  Among so many, a few crests are flat to within the difference between
    # this f32 pipeline and the checker's f64 transforms (< 1e-6), and which of two neighbouring samples is
    # the maximum is then not defined by the input: positions may differ by a few samples there, everything
    # else must agree.
    p2 = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=0.0, prominence=0.02).params(sr, gpu.Scale.LIB)
    exp2 = oracle.calc_chunks(sr, hay, needle, p2.chunk, p2.overlap, 0.02, p2.min_distance, 0.0, cap=1 << 20)
    got2 = algo.match_device(hbuf.ptr, h, p2, cap=1 << 20)
    assert len(exp2) < (1 << 20) and abs(len(got2) - len(exp2)) <= max(2, len(exp2) // 2000)
    gpos = np.array([g.start for g in got2], dtype=np.int64)
    epos = np.array([e[0] for e in exp2], dtype=np.int64)
    ehgt = np.array([e[2] for e in exp2])
    near = np.clip(np.searchsorted(epos, gpos), 1, len(epos) - 1)
    pick_left = np.abs(epos[near - 1] - gpos) <= np.abs(epos[near] - gpos)
    idx = np.where(pick_left, near - 1, near)
    ghgt = np.array([g.height for g in got2])
    close = (np.abs(epos[idx] - gpos) <= 16) & (np.abs(ehgt[idx] - ghgt) < TOL)
    exact = epos[idx] == gpos
    assert close.mean() > 0.999 and exact.mean() > 0.99, (close.mean(), exact.mean(), len(got2), len(exp2))


# ---------------------------------------------------------------------------
# N = 2^23 = 1024 x 8192
# ---------------------------------------------------------------------------
def test_plan_2_23_correlate_and_match(gpu, oracle):
    """The 1024-row column kernels (N = 2^23): several blocks and pairs with an odd block count through
    level 1 against the checker, and calc_chunks (fused score scan of the 1024-row K3, sparse raw scores,
    i16 ingest) against the checker and against the default plan's offsets."""
    rng = np.random.default_rng(23)
    needle = rng.uniform(-1, 1, 100000).astype(np.float32)
    within = rng.uniform(-1, 1, 20_000_000).astype(np.float32)
    algo = gpu.HipConvolve(needle)
    algo.set_option("log_n", 23)
    got = algo.correlate_with_sample(within, gpu.Mode.Valid, True)
    expect = oracle.correlate(within, needle, oracle.MODE_VALID, oracle.SCALE_LIB, oracle.FFT_POW2)
    assert got.shape == expect.shape and float(np.abs(got - expect).max()) < TOL
    del got, expect, within
    sr = 8000
    s = 5 * sr
    nd = oracle.synth_uniform(23, 0, 0, s)
    hay = oracle.synth_uniform(23, 1, 0, 2700 * sr)             # 21.6 M samples: three blocks of 2^23
    plants = [17.0, 1043.5, 1044.75, 2099.0, 2690.0]
    for t in plants:
        off = int(t * sr)
        hay[off:off + s] += nd
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=5.0, distance_s=30.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, nd, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    wide = gpu.HipConvolve(nd)
    wide.set_option("log_n", 23)
    for _ in range(2):
        assert_same(wide.match(hay, p), exp)
    assert pos(gpu.HipConvolve(nd).match(hay, p)) == pos(wide.match(hay, p))
    bufs = [gpu.DeviceBuffer.from_numpy(0, hay) for _ in range(2)]
    res = wide.match_batch_device([b.ptr for b in bufs], [hay.size] * 2, p)
    assert key(res[0]) == key(res[1]) == key(wide.match(hay, p))
    gpu.set_option("dense_scores", 1)
    try:
        assert key(wide.match(hay, p)) == key(res[0])
    finally:
        gpu.set_option("dense_scores", 0)


# ---------------------------------------------------------------------------
# certificate failures inside a batch: redone on the device
# ---------------------------------------------------------------------------
def test_failed_certificates_in_a_batch_are_redone_on_the_device(gpu, oracle):
    """Chunks whose minimum lies far below what the K3 tiles sampled (inverted copies of the needle: dips to -1 a few
    scores wide) fail their certificate.  In a batch the pick marks the block pairs that feed them, K3 runs again
    for those pairs with every run written and the chunks are picked again, all on the device -- once a failure
    has armed that path for the needle; the results equal the single calls' (which redo such chunks from the
    host) and the checker's, bit for bit, call after call (the write threshold's history moves in between),
    also with a dip in every chunk."""
    sr = 44100
    s = 3 * sr
    needle = oracle.synth_uniform(95, 0, 0, s)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=3.0, distance_s=2.0, prominence=0.13).params(sr, gpu.Scale.LIB)

    def make(seed, dips, hits):
        hay = oracle.synth_uniform(95, seed, 0, 400 * sr)
        for t in dips:
            hay[int(t * sr):int(t * sr) + s] -= needle
        for t in hits:
            hay[int(t * sr):int(t * sr) + s] += needle
        return hay

    hays = [make(1, (70.0,), (75.0, 150.0)),                              # one chunk with a dip
            make(2, (), (20.0, 333.3)),                                     # none
            make(3, (10.0, 70.0, 130.0, 190.0, 250.0, 310.0, 370.0), (45.0, 200.0)),   # a dip in every chunk
            make(4, (199.0, 301.0), (203.5, 305.0))]
    exps = [oracle.calc_chunks(sr, h, needle, p.chunk, p.overlap, 0.13, p.min_distance, 2.0) for h in hays]
    assert [len(e) for e in exps] == [2, 2, 2, 2]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
    ptrs, lens = [b.ptr for b in bufs], [h.size for h in hays]
    single = gpu.HipConvolve(needle)
    want = [key(single.match_device(b.ptr, n, p)) for b, n in zip(bufs, lens)]
    for got, exp in zip(want, exps):
        assert [q[0] for q in got] == [e[0] for e in exp]
    algo = gpu.HipConvolve(needle)
    gpu.set_option("profile_mask", -1)
    # the first call finds the failures and redoes those chunks from the host; that arms the device-side redo for
    # the needle's next haystacks (a batch that never sees a failure does not pay for the redo's launches)
    with gpu.Profile(0) as prof:
        res = algo.match_batch_device(ptrs, lens, p)
        assert [key(r) for r in res] == want
        first_k3 = prof.query("k3_cols_inv")[1]
    assert first_k3 > len(hays)
    with gpu.Profile(0) as prof:
        for _ in range(3):
            res = algo.match_batch_device(ptrs, lens, p)
            assert [key(r) for r in res] == want
        k3_launches, redo_launches = prof.query("k3_cols_inv")[1], prof.query("other")[1]
    # one K3 per haystack and call -- a host-side redo would launch more -- and one (mostly empty) device-side redo launch
    assert k3_launches == 3 * len(hays) and redo_launches == 3 * len(hays)
    res = algo.match_batch_device(ptrs[::-1], lens[::-1], p)
    assert [key(r) for r in res] == want[::-1]
    for r, e in zip(res[::-1], exps):
        assert_same(r, e)
    # a long batch through a FRESH handle: nothing is armed when the call starts, the first failures take the host
    # path, and the redo may get armed while the call is still queuing (whenever a flag has reached host memory by
    # then) -- whichever haystack takes which path, the results are the single calls'
    fresh = gpu.HipConvolve(needle)
    order = [0, 2, 1, 3] * 4
    res = fresh.match_batch_device([ptrs[i] for i in order], [lens[i] for i in order], p)
    assert [key(r) for r in res] == [want[i] for i in order]
    # i16 input and the half-precision levels go the same way
    pool = gpu.Pool(needle, [0, 0])
    assert [key(r) for r in pool.match_batch_device(ptrs, lens, p)] == want
    pool.close()
