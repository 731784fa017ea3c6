"""GPU tests of the round-4 work, all through the C ABI:
one long haystack split over the slots of a pool (am_long_plan / am_match_part_device / am_merge_peaks /
am_pool_match_long*), the scratch buffers of a ragged batch (option debug_no_realloc), the
device / host redo paths of failed certificates chosen deterministically (option debug_redo_arm_at), and the
streaming-ingest corner cases of the round-3 review."""
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


def assert_close_peaks(got, one, tol=TOL):
    assert pos(got) == pos(one)
    for g, o in zip(got, one):
        assert abs(g.height - o.height) < tol and abs(g.prominence - o.prominence) < tol


# ---------------------------------------------------------------------------
# one long haystack over several slots
# ---------------------------------------------------------------------------
def long_case(oracle, sr, minutes, s, chunk_s, seed=71):
    """A long haystack with hits that straddle every possible cut (a window boundary i * chunk): a hit
    that starts just before a boundary, one just behind it, a pair closer together than the overshadow
    distance on either side of a boundary (the weaker one must go, whichever part found it)."""
    needle = oracle.synth_uniform(seed, 0, 0, s)
    n = minutes * 60 * sr
    hay = oracle.synth_uniform(seed, 1, 0, n)
    chunk = int(chunk_s * sr)
    plants = []
    nwin = n // chunk
    for i in range(1, nwin):
        if i % 3 == 0:
            plants.append((i * chunk - s // 2, 1.0))          # straddles the boundary
        elif i % 3 == 1:
            plants.append((i * chunk - s - 7, 1.0))           # ends just before it
            plants.append((i * chunk + 11, 0.7))              # a weaker one just behind it: overshadowed
        else:
            plants.append((i * chunk + 5, 1.0))
    plants.append((n - s - 3, 1.0))                           # three scores before the end of the score array
    for off, g in plants:
        hay[off:off + s] += np.float32(g) * needle
    return needle, hay, plants


@pytest.mark.parametrize("fmt", ["f32", "s16"])
def test_long_haystack_over_pool_slots_equals_am_match(gpu, oracle, fmt):
    """am_pool_match_long / _device with 1, 2 and 3 slots on device 0 (audio_matcher.rs:104-140: the windows
    of one haystack fanned out, ONE sort + overshadow pass over the union): offsets and plateau ends equal
    am_match on the whole buffer and the checker, heights and prominences within 1e-4; hits straddle every
    cut, a weaker hit next to a cut is overshadowed by its neighbour from the other part."""
    sr = 16000
    s = 2 * sr
    needle, hay, plants = long_case(oracle, sr, 40, s, 60.0)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=2.0, distance_s=5.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    if fmt == "s16":
        # the same signal as interleaved i16 stereo frames (left = right, so that the down-mix gives it back scaled)
        q = np.clip(np.rint(hay * 20000.0), -32768, 32767).astype(np.int16)
        data = np.repeat(q, 2)
        nq = np.clip(np.rint(needle * 20000.0), -32768, 32767).astype(np.int16)
        algo = gpu.HipConvolve.from_pcm16(np.repeat(nq, 2))
        whole = algo.match_pcm16(data, p)
        pool_needle = gpu.pcm_s16_stereo_to_mono(np.repeat(nq, 2))
        gfmt = gpu.Fmt.S16_STEREO
    else:
        data = hay
        algo = gpu.HipConvolve(needle)
        whole = algo.match(hay, p)
        pool_needle = needle
        gfmt = gpu.Fmt.F32_MONO
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
        assert_same(whole, exp)
    strong = sorted(off for off, g in plants if g == 1.0)
    assert [q.start for q in whole] == strong            # every weak neighbour is overshadowed, every strong hit found
    n = hay.size
    for slots in (1, 2, 3):
        pool = gpu.Pool(pool_needle, [0] * slots)
        got = pool.match_long(data, p, fmt=gfmt)
        assert_close_peaks(got, whole)
        # resident parts: the plan is a pure function, the parts are views into one resident copy
        buf = gpu.DeviceBuffer.from_numpy(0, data)
        plan = [gpu.long_plan(n, s, p, slots, i) for i in range(slots)]
        assert sum(pl[1] for pl in plan) == (n - s) // p.chunk + 1
        assert all(plan[i][0] + plan[i][1] == plan[i + 1][0] for i in range(slots - 1))
        got_dev = pool.match_long_device([buf.ptr + 4 * pl[2] for pl in plan], n, p, fmt=gfmt)
        assert key(got_dev) == key(got)
        # the building blocks on their own: parts, concatenated in part order, then ONE merge
        raw = []
        for pl in plan:
            raw += gpu.match_part_device(algo, buf.ptr + 4 * pl[2], pl[3], p, pl[1], pl[2], fmt=gfmt)
        assert len(raw) > len(whole)                  # the weak neighbours are still in
        assert_close_peaks(gpu.merge_peaks(p, raw), whole)
        pool.close()
        buf.free()


def test_long_haystack_edge_cases(gpu, oracle):
    """More slots than windows, a haystack shorter than the needle, a short tail window, MyConvolve scaling
    (every window scaled by its own length, audio_matcher.rs:442-448), a NaN next to a cut (the window that holds
    it yields nothing, its neighbours across the cut are untouched: audio_matcher.rs:114-122), progress events with
    the whole haystack's chunk indices, capacity."""
    sr = 8000
    s = sr
    needle = oracle.synth_uniform(72, 0, 0, s)
    hay = oracle.synth_uniform(72, 1, 0, 250 * sr + 4321)
    for t in (10.0, 55.0, 60.2, 119.9, 200.0, 248.0):
        off = int(t * sr)
        hay[off:off + s] += needle
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=1.0, distance_s=3.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    whole = algo.match(hay, p)
    assert len(whole) == 6
    pool7 = gpu.Pool(needle, [0] * 7)                     # five windows, seven slots
    assert_close_peaks(pool7.match_long(hay, p), whole)
    assert pool7.match_long(hay[:s - 1], p) == []
    assert pool7.match_long(np.zeros(0, np.float32), p) == []
    assert [gpu.long_plan(s - 1, s, p, 3, i)[1] for i in range(3)] == [0, 0, 0]
    pool7.close()
    pool2 = gpu.Pool(needle, [0, 0])
    pm = gpu.Config(chunk_size_s=60.0, overlap_length_s=1.0, distance_s=3.0, prominence=1e-7).params(sr, gpu.Scale.MY)
    one = algo.match(hay, pm)
    two = pool2.match_long(hay, pm)
    assert pos(two) == pos(one)
    for g, o in zip(two, one):
        assert abs(g.height - o.height) <= 1e-4 * abs(o.height) and abs(g.prominence - o.prominence) <= 1e-4 * abs(o.prominence)
    # a NaN 100 samples behind the cut between the parts (5 windows over 2 slots: the cut is at window 2 = 120 s)
    bad = hay.copy()
    bad[120 * sr + 100] = np.nan
    exp_bad = oracle.calc_chunks(sr, bad, needle, p.chunk, p.overlap, 0.13, p.min_distance, 3.0)
    one_bad = algo.match(bad, p)
    assert_same(one_bad, exp_bad)
    assert len(one_bad) < len(whole)                    # the hit at 119.9 s reaches into the poisoned window ...
    assert_close_peaks(pool2.match_long(bad, p), one_bad)
    # progress: haystack 0, chunk indices 0..4 of 5, every chunk stage 0 before stage 1
    events, hay_events = [], []
    gpu.set_chunk_progress_callback(lambda k, i, n, stage: events.append((k, i, n, stage)))
    gpu.set_progress_callback(lambda k, stage, n: hay_events.append((k, stage, n)))
    try:
        pool2.match_long(hay, p)
    finally:
        gpu.set_chunk_progress_callback(None)
        gpu.set_progress_callback(None)
    assert hay_events == [(0, 0, 5), (0, 1, 5)]
    assert sorted(events) == sorted((0, i, 5, st) for i in range(5) for st in (0, 1))
    for i in range(5):
        assert events.index((0, i, 5, 0)) < events.index((0, i, 5, 1))
    # capacity: the count is reported
    a = np.ascontiguousarray(hay)
    cnt = C.c_size_t(0)
    rc = gpu.lib().am_pool_match_long(pool2._p, a.ctypes.data, a.size, 0, C.byref(p), None, 0, C.byref(cnt))
    assert rc == gpu.AM_ERR_CAPACITY and cnt.value == 6
    rc = gpu.lib().am_pool_match_long(pool2._p, a.ctypes.data, a.size, 9, C.byref(p), None, 0, C.byref(cnt))
    assert rc == gpu.AM_ERR_INVALID_ARG
    pool2.close()


def test_pool_device_entry_points_check_where_a_pointer_lives(gpu, oracle):
    """am_pool_match_batch_device / am_pool_match_long_device ask the runtime where each pointer lives
    (hipPointerGetAttributes) instead of trusting the caller: host memory is refused by name."""
    sr = 8000
    needle = oracle.synth_uniform(73, 0, 0, sr)
    hay = oracle.synth_uniform(73, 1, 0, 30 * sr)
    p = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0, distance_s=3.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    pool = gpu.Pool(needle, [0])
    with pytest.raises(gpu.AudioMatchError) as e:
        pool.match_batch_device([hay.ctypes.data], [hay.size], p)
    assert e.value.code == gpu.AM_ERR_INVALID_ARG and "haystack 0" in str(e.value)
    with pytest.raises(gpu.AudioMatchError) as e:
        pool.match_long_device([hay.ctypes.data], hay.size, p)
    assert e.value.code == gpu.AM_ERR_INVALID_ARG
    buf = gpu.DeviceBuffer.from_numpy(0, hay)
    assert pool.match_batch_device([buf.ptr], [hay.size], p) == [[]]
    pool.close()


# ---------------------------------------------------------------------------
# scratch buffers of a ragged batch; the two redo paths, deterministically
# ---------------------------------------------------------------------------
def dip_case(oracle, sr, s, secs, seed, dips, hits, needle):
    hay = oracle.synth_uniform(95, seed, 0, secs * sr)
    for t in dips:
        hay[int(t * sr):int(t * sr) + s] -= needle
    for t in hits:
        hay[int(t * sr):int(t * sr) + s] += needle
    return hay


def test_ascending_ragged_batch_allocates_nothing_while_queueing(gpu, oracle):
    """A batch whose haystacks get LONGER (60 s, 10 min, 1 h; certificate failures in the last) with the option
    debug_no_realloc: every scratch buffer -- work matrices, level-0 summaries, ballots and thresholds, redo flags --
    is sized for the longest haystack before the first kernel is queued; an ensure() that still had to allocate
    inside the queueing loop would fail the call (round 3 lost a peak to a buffer re-allocated under a running pick,
    gpurun_out/r03q).  Fresh contexts' buffers start empty, so am_shutdown first; results == single calls."""
    sr = 44100
    s = 3 * sr
    needle = oracle.synth_uniform(95, 0, 0, s)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=3.0, distance_s=2.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    hays = [dip_case(oracle, sr, s, 60, 1, (), (20.0,), needle),
            dip_case(oracle, sr, s, 600, 2, (), (100.0, 500.0), needle),
            dip_case(oracle, sr, s, 3600, 3, (70.0, 1990.0), (75.0, 2000.0, 3590.0), needle)]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
    ptrs, lens = [b.ptr for b in bufs], [h.size for h in hays]
    single = gpu.HipConvolve(needle)
    want = [key(single.match_device(b.ptr, n, p)) for b, n in zip(bufs, lens)]
    assert [len(w) for w in want] == [1, 2, 3]
    single.close()
    assert gpu.lib().am_shutdown() == 0                  # every scratch buffer of the context is gone
    algo = gpu.HipConvolve(needle)
    gpu.set_option("debug_no_realloc", 1)
    try:
        for arm in (-2, 0, -1):                          # natural arming, device redo from the start, host redo only
            gpu.set_option("debug_redo_arm_at", arm)
            res = algo.match_batch_device(ptrs, lens, p)
            assert [key(r) for r in res] == want, arm
        # descending and mixed orders, and the several-needle engine, under the same rule
        res = algo.match_batch_device(ptrs[::-1], lens[::-1], p)
        assert [key(r) for r in res] == want[::-1]
        other = gpu.HipConvolve(oracle.synth_uniform(96, 0, 0, s))
        multi = gpu.match_multi_batch_device([algo, other], ptrs, lens, p)
        assert [pos(m[0]) for m in multi] == [[(q[0], q[1]) for q in w] for w in want]
        assert all(m[1] == [] for m in multi)
    finally:
        gpu.set_option("debug_no_realloc", 0)
        gpu.set_option("debug_redo_arm_at", -2)
    assert gpu.get_option("debug_no_realloc") == 0 and gpu.get_option("debug_redo_arm_at") == -2


def test_redo_paths_chosen_deterministically(gpu, oracle):
    """Which path a failed chunk takes in a batch -- redone on the device beside the next haystack's transforms, or
    from the host after the call -- normally depends on when the first failure flag reaches host memory.  The
    option debug_redo_arm_at pins it: -1 = host path for every haystack, 0 = device path from the start, k = the
    switch-over at haystack k.  Every choice gives the single calls' results bit for bit, and the launch counts
    show that the chosen path really ran."""
    sr = 44100
    s = 3 * sr
    needle = oracle.synth_uniform(95, 0, 0, s)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=3.0, distance_s=2.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    hays = [dip_case(oracle, sr, s, 400, 1, (70.0,), (75.0, 150.0), needle),
            dip_case(oracle, sr, s, 400, 2, (), (20.0, 333.3), needle),
            dip_case(oracle, sr, s, 400, 3, (10.0, 70.0, 130.0, 190.0, 250.0, 310.0, 370.0), (45.0, 200.0), needle),
            dip_case(oracle, sr, s, 400, 4, (199.0, 301.0), (203.5, 305.0), needle)]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
    order = [0, 2, 1, 3, 2, 0]
    ptrs, lens = [bufs[i].ptr for i in order], [hays[i].size for i in order]
    single = gpu.HipConvolve(needle)
    want = [key(single.match_device(bufs[i].ptr, hays[i].size, p)) for i in order]
    exp = oracle.calc_chunks(sr, hays[2], needle, p.chunk, p.overlap, 0.13, p.min_distance, 2.0)
    assert [q[0] for q in want[1]] == [e[0] for e in exp]
    n = len(order)
    gpu.set_option("profile_mask", -1)
    try:
        counts = {}
        for arm in (-1, 0, 3, n):
            gpu.set_option("debug_redo_arm_at", arm)
            algo = gpu.HipConvolve(needle)                      # a fresh handle: no history of failures
            with gpu.Profile(0) as prof:
                res = algo.match_batch_device(ptrs, lens, p)
                counts[arm] = (prof.query("k3_cols_inv")[1], prof.query("other")[1])
            assert [key(r) for r in res] == want, arm
            algo.close()
        # "other" counts the device-side redo launches (one per armed haystack; nothing else of a warm context lands there)
        assert counts[0][1] - counts[-1][1] == n and counts[3][1] - counts[-1][1] == n - 3 and counts[n][1] == counts[-1][1]
        # the host path launches K3 again per failed chunk: the fewer haystacks are armed, the more K3 launches
        assert counts[0][0] == n and counts[-1][0] > counts[3][0] > counts[0][0]
    finally:
        gpu.set_option("debug_redo_arm_at", -2)


# ---------------------------------------------------------------------------
# streaming ingest: the layout of the side buffer, the rule the early pairs were written under
# ---------------------------------------------------------------------------
def push_pieces(stream, data, piece):
    for off in range(0, data.size, piece):
        stream.push(data[off:off + piece])


def test_stream_announced_length_with_more_blocks_than_the_real_one(gpu, oracle):
    """The stream lays its buffers out for the ANNOUNCED length (plus slack); the thresholds K3 records sit behind
    the ballots, at an offset that depends on the block count.  A file that ends up with FEWER blocks than the
    layout of its early pairs must keep that layout in the final pass (round-3 review: the certificate read the
    early blocks' thresholds at the wrong offset) -- announced 1.5 hops too long, exactly announced where the
    slack alone adds a block, and the stream reused for a second file of a different length.  Dips (inverted
    needles) in the early blocks make the certificate fail if it reads a wrong threshold silently passing, so the
    results are compared with am_match bit for bit, with and without dense scores."""
    sr = 8000
    s = 2 * sr
    needle = oracle.synth_uniform(81, 0, 0, s)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=2.0, distance_s=30.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    hop = (2 ** 21 - s + 1) // 1024 * 1024
    n = 7 * hop - 100000 + s - 1                                  # 7 blocks of the 2^21 plan; the stream's slack alone makes it 8

    def make(length, seed):
        hay = oracle.synth_uniform(81, seed, 0, length)
        for t in (17.0, 300.5, 700.0, 1100.0):
            off = int(t * sr)
            if off + s <= length:
                hay[off:off + s] += needle
        for t in (100.0, 640.0):                                   # dips to -1: those chunks need every run written
            off = int(t * sr)
            if off + s <= length:
                hay[off:off + s] -= needle
        return hay

    hay = make(n, 1)
    want = key(algo.match(hay, p))
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    assert [q[0] for q in want] == [e[0] for e in exp] and len(want) >= 3
    gpu.set_option("profile_mask", -1)
    for announced in (n + hop + hop // 2, n, n + 3 * hop):
        st = gpu.MatchStream(algo, p, announced)
        with gpu.Profile(0) as prof:
            push_pieces(st, hay, 3000000)
            early = prof.query("k1_cols_fwd")[1]
            got = key(st.finish())
        assert early >= 1, "no block pair was transformed while the samples arrived"
        assert got == want, announced
        shorter = make(n - 2 * hop - 777, 2)                      # reuse: a second file with another block count
        push_pieces(st, shorter, 2500000)
        assert key(st.finish()) == key(algo.match(shorter, p)), announced
        st.close()
    # the write rule changes between push and finish: the early pairs (sparse) must not be mixed with a dense pick
    st = gpu.MatchStream(algo, p, n)
    push_pieces(st, hay[:4 * hop], 3000000)
    gpu.set_option("dense_scores", 1)
    try:
        push_pieces(st, hay[4 * hop:], 3000000)
        assert key(st.finish()) == want
    finally:
        gpu.set_option("dense_scores", 0)
    # ... and the other way round (early pairs dense, the rest sparse)
    gpu.set_option("dense_scores", 1)
    try:
        push_pieces(st, hay[:4 * hop], 3000000)
    finally:
        gpu.set_option("dense_scores", 0)
    push_pieces(st, hay[4 * hop:], 3000000)
    assert key(st.finish()) == want
    # ... and a change that only finish sees (every early pair written densely, the pick sparse)
    gpu.set_option("dense_scores", 1)
    try:
        push_pieces(st, hay, 3000000)
    finally:
        gpu.set_option("dense_scores", 0)
    assert key(st.finish()) == want
    # the stream is as good as new afterwards: early pairs run again for the next file
    with gpu.Profile(0) as prof:
        push_pieces(st, hay, 3000000)
        early = prof.query("k1_cols_fwd")[1]
        assert key(st.finish()) == want and early >= 1
    st.close()


def test_accumulating_k3_takes_any_score_pointer(gpu, oracle):
    """A partitioned needle (longer than 2^22 samples: K3 adds the segments' partial sums into the score array)
    into a caller buffer that is only 4-byte aligned: MyConvolve::correlate has no alignment rule
    (audio_matcher.rs:414-457); the sums equal the aligned call's bit for bit."""
    s = (1 << 22) + 12345
    w = s + 70001
    needle = oracle.synth_uniform(83, 0, 0, s)
    within = oracle.synth_uniform(83, 1, 0, w)
    algo = gpu.HipConvolve(needle)
    d_in = gpu.DeviceBuffer.from_numpy(0, within)
    n_out = w - s + 1
    d_out = gpu.DeviceBuffer(0, 4 * (n_out + 4))
    res = {}
    for shift in (0, 1):
        got = C.c_size_t(0)
        gpu._check(gpu.lib().am_correlate_device(algo._h, d_in.ptr, w, int(gpu.Mode.Valid), int(gpu.Scale.LIB),
                                                  d_out.ptr + 4 * shift, n_out, C.byref(got)))
        assert got.value == n_out
        res[shift] = d_out.to_numpy(np.float32, n_out + 4)[shift:shift + n_out].copy()
    assert np.array_equal(res[0], res[1])
    idx = np.array([0, 1, 2, 777, n_out // 2, n_out - 2, n_out - 1])
    inv = 1.0 / float(np.dot(needle.astype(np.float64), needle.astype(np.float64)))
    ref = np.array([np.dot(within[j:j + s].astype(np.float64), needle.astype(np.float64)) * inv for j in idx])
    assert np.abs(res[1][idx] - ref).max() < TOL


# ---------------------------------------------------------------------------
# host feed: staging ring of the stream, pinned host buffers
# ---------------------------------------------------------------------------
def test_stream_decoder_sized_pushes_and_pinned_buffers(gpu, oracle):
    """am_match_stream_push with the decoder's piece size (1152 frames, mp3_reader.rs:28-37): every piece is a host
    memcpy into the two-slot pinned staging ring, full slots leave asynchronously, pieces of 1 MB and more bypass the
    ring -- in any mixture the result is am_match's, bit for bit, the block pairs are still transformed while the
    samples arrive, and the stream is reusable.  The same from pinned host memory (am_host_alloc, am_host_register),
    also through the pool's copier threads."""
    sr = 8000
    s = 2 * sr
    needle = oracle.synth_uniform(61, 0, 0, s)
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=2.0, distance_s=30.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    hay = oracle.synth_uniform(62, 1, 0, 1150 * sr)
    for t in (17.0, 300.5, 519.9, 1000.0, 1147.5):
        off = int(t * sr)
        hay[off:off + s] += needle
    want = key(algo.match(hay, p))
    assert [q[0] for q in want] == [int(t * sr) for t in (17.0, 300.5, 519.9, 1000.0, 1147.5)]
    gpu.set_option("profile_mask", -1)
    st = gpu.MatchStream(algo, p, hay.size)
    with gpu.Profile(0) as prof:
        push_pieces(st, hay, 1152)                                   # 8000 decoder-sized pieces
        early = prof.query("k1_cols_fwd")[1]
        assert key(st.finish()) == want
    assert early >= 1, "no block pair was transformed while the samples arrived"
    # a mixture: small pieces that straddle slot boundaries, pieces just below and at the direct-copy threshold, large ones
    sizes = [1152] * 2000 + [300000, 1, 4607] + [4607] * 400 + [262143, 262144, 1 << 20, 5] + [99999] * 30 + [3000000]
    off = 0
    for k in sizes:
        st.push(hay[off:off + k])
        off += k
    push_pieces(st, hay[off:], 1152)
    assert key(st.finish()) == want
    assert st.finish() == []
    # the length unknown in advance: the device buffer grows while the ring holds samples that have not left yet
    st2 = gpu.MatchStream(algo, p, 0)
    push_pieces(st2, hay, 50000)
    assert key(st2.finish()) == want
    st2.close()
    # pinned host memory: allocated by the library, or registered for the duration
    pin = gpu.PinnedArray(hay.shape, np.float32)
    pin.array[:] = hay
    push_pieces(st, pin.array, 8 << 20)
    assert key(st.finish()) == want
    assert key(algo.match(pin.array, p)) == want
    with gpu.registered(hay) as h:
        push_pieces(st, h, 8 << 20)
        assert key(st.finish()) == want
    st.close()
    pool = gpu.Pool(needle, [0, 0])
    res = pool.match_batch([pin.array, hay, pin.array], p)
    assert [key(r) for r in res] == [want] * 3
    assert_close_peaks(pool.match_long(pin.array, p), algo.match(hay, p))
    pool.close()
    pin.free()
    assert gpu.lib().am_host_alloc(0, None) == gpu.AM_ERR_INVALID_ARG
    assert gpu.lib().am_host_register(None, 16) == gpu.AM_ERR_INVALID_ARG


def test_matrix_core_row_kernel_is_result_equivalent(gpu, oracle):
    """Option k2_mfma (an A/B experiment, off by default): the half-precision row kernel with its 16- and 32-point
    butterflies as v_mfma_f32_16x16x32_f16 products.  Same offsets as the checker, scores within the half pipeline's
    1e-3, on both register plans (2^21: 5 s needle, 2^22: 8 s needle) and through a second work matrix (several needles)."""
    sr = 44100
    gpu.set_option("k2_mfma", 1)
    try:
        for secs, seed in ((5.0, 91), (8.0, 92)):
            s = int(secs * sr)
            needles = [oracle.synth_uniform(seed, 300 + k, 0, s) for k in range(2)]
            hay = oracle.synth_uniform(seed, 1, 0, 130 * sr)
            plants = [[int(12.25 * sr), int(91.0 * sr)], [int(47.5 * sr)]]
            for n_, offs in zip(needles, plants):
                for off in offs:
                    hay[off:off + s] += n_
            p = gpu.Config(chunk_size_s=60.0, overlap_length_s=secs, distance_s=20.0, prominence=0.13).params(sr, gpu.Scale.LIB)
            algos = [gpu.HipConvolve(n_) for n_ in needles]
            for a in algos:
                a.set_option("half_pipeline", 2)
            buf = gpu.DeviceBuffer.from_numpy(0, hay)
            exps = [oracle.calc_chunks(sr, hay, n_, p.chunk, p.overlap, 0.13, p.min_distance, 20.0) for n_ in needles]
            assert [[e[0] for e in ex] for ex in exps] == plants
            for _ in range(2):
                for a, ex in zip(algos, exps):
                    assert_same(a.match_device(buf.ptr, hay.size, p), ex, tol=1e-3)
            for got, ex in zip(gpu.match_multi_device(algos, buf.ptr, hay.size, p), exps):
                assert_same(got, ex, tol=1e-3)
            # the whole score vector of one window against the checker
            win = hay[:3 * 1024 * 1024]
            sc = algos[0].correlate_with_sample(win, gpu.Mode.Valid, True)
            ref = oracle.correlate(win, needles[0], oracle.MODE_VALID, oracle.SCALE_LIB)
            assert np.abs(sc - ref).max() < 1e-3
    finally:
        gpu.set_option("k2_mfma", 0)
    assert gpu.get_option("k2_mfma") == 0


# ---------------------------------------------------------------------------
# the odd last block on the smaller plan (option tail_block, on by default)
# ---------------------------------------------------------------------------
def tail_geometry(s):
    """Hops of the 2^22 plan and of the 2^21 plan for a needle of s samples (am_api.hip, plan_geometry)."""
    hop = ((1 << 22) - s + 1) // 1024 * 1024
    hop_t = ((1 << 21) - s + 1) // 1024 * 1024
    return hop, hop_t


def tail_case(oracle, sr, s, rest, seed, where, even_blocks=2):
    """A haystack of `even_blocks` blocks of the 2^22 plan + `rest` scores (an odd last block: with the option on, scores
    [T, T + rest) come from one pair of the 2^21 plan), hits planted at the score positions where(T, hop_t, out)."""
    hop, hop_t = tail_geometry(s)
    T = even_blocks * hop
    out = T + rest
    needle = oracle.synth_uniform(seed, 0, 0, s)
    hay = oracle.synth_uniform(seed, 1, 0, out + s - 1)
    offs = sorted(where(T, hop_t, out))
    for off in offs:
        hay[off:off + s] += needle
    return needle, hay, offs, T


def test_odd_last_block_on_the_smaller_plan(gpu, oracle):
    """10 s needle (the 2^22 plan), haystacks with an odd, part-filled last block: with tail_block = 1 (default) the main
    pass stops at the even block boundary T and the scores behind it come from one pair of the 2^21 plan, every run
    written -- beside the main pass for a single haystack, several haystacks' tails per launch in a batch.  Hits right
    before T, on the first tail scores, on either side of the tail's own block boundary and on the last scores:
    offsets == checker, heights within 1e-4, the same with the option off (the last block as half of a full pair);
    the launch counts (the tail's kernels are profiled as "other") say which form ran; and a haystack's results are
    the same bit for bit alone and in any batch, although a batch computes its tails several per launch."""
    sr = 44100
    s = 10 * sr
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=2.0, prominence=0.3).params(sr, gpu.Scale.LIB)
    cases = [tail_case(oracle, sr, s, 1900000, 201, lambda T, ht, out: [sr + 7, T - 5, T + ht - 2, out - 3]),   # two tail blocks
             tail_case(oracle, sr, s, 1900000, 202, lambda T, ht, out: [T + 3, T + ht + 1, out - 3]),
             tail_case(oracle, sr, s, 500000, 203, lambda T, ht, out: [T - 3 * sr, T, out - 2]),               # one tail block
             tail_case(oracle, sr, s, 2 * tail_geometry(s)[1], 204, lambda T, ht, out: [T + ht - 1, out - 2]),  # the most that fits
             tail_case(oracle, sr, s, 1900000, 206, lambda T, ht, out: [T - 2 * sr, T + 7, out - 3], even_blocks=4)]
    even = oracle.synth_uniform(205, 1, 0, 3 * tail_geometry(s)[0] + 1000 + s)                              # four blocks: no tail
    gpu.set_option("profile_mask", -1)
    algo = gpu.HipConvolve(cases[0][0])
    try:
        for needle, hay, offs, T in cases:
            a = gpu.HipConvolve(needle)
            exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.3, p.min_distance, 2.0)
            assert [e[0] for e in exp] == offs
            buf = gpu.DeviceBuffer.from_numpy(0, hay)
            res, launches = {}, {}
            for on in (1, 0):
                gpu.set_option("tail_block", on)
                a.match_device(buf.ptr, hay.size, p)            # (plans and spectra exist from here on)
                with gpu.Profile(0) as prof:
                    res[on] = a.match_device(buf.ptr, hay.size, p)
                    launches[on] = (prof.query("k1_cols_fwd")[1], prof.query("other")[1])
                assert_same(res[on], exp)
            assert launches[1][0] == launches[0][0] == 1 and launches[1][1] == launches[0][1] + 3
            assert_close_peaks(res[1], res[0], tol=1e-5)
            a.close()
        # bit-identical alone and in a batch (same needle for all: the first case's), and the batch's tails really
        # went several per launch: three launches and one commit per haystack with a tail, not three per haystack
        gpu.set_option("tail_block", 1)
        hays = [c[1] for c in cases[:2]] + [even, cases[2][1], cases[4][1]]
        bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
        want = [key(algo.match_device(b.ptr, h.size, p)) for b, h in zip(bufs, hays)]
        for order in ([0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [0, 0, 2, 0, 3, 3, 1, 4, 4, 1, 0, 3, 1]):
            ptrs, lens = [bufs[i].ptr for i in order], [hays[i].size for i in order]
            algo.match_batch_device(ptrs, lens, p)
            with gpu.Profile(0) as prof:
                got = algo.match_batch_device(ptrs, lens, p)
                other = prof.query("other")[1]
            assert [key(r) for r in got] == [want[i] for i in order], order
            with_tail = sum(1 for i in order if i != 2)
            assert other == 3 * -(-with_tail // 8) + with_tail, (order, other)
    finally:
        gpu.set_option("tail_block", 1)
        algo.close()
    assert gpu.get_option("tail_block") == 1


def test_odd_last_block_redo_nan_and_half_paths(gpu, oracle):
    """The rare paths around the tail block: (a) a chunk that reaches into the tail fails its certificate (a dip
    deeper than half a prominence) -- redone from the host in a single call (the redo recomputes the tail as well: the
    score buffers have moved on) and on the device in a batch; (b) a NaN sample that only the tail's transform reads:
    its window yields nothing, the clean window whose last scores came from the poisoned tail pair is correlated again
    on its own -- alone and in a batch; (c) the packed-f16 pipeline on i16 stereo frames: same offsets with the option
    on and off, alone and in a batch."""
    sr = 44100
    s = 10 * sr
    hop, hop_t = tail_geometry(s)
    T = 2 * hop
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=2.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    needle, hay, offs, _ = tail_case(oracle, sr, s, 1900000, 211, lambda T, ht, out: [20 * sr, T + 3 * sr, out - 1 - sr])
    assert gpu.get_option("tail_block") == 1
    # (a) dips in the chunk that holds T and in the last chunk
    dipped = hay.copy()
    for off in (T - 4 * sr - s, T + 12 * sr):
        dipped[off:off + s] -= needle
    exp = oracle.calc_chunks(sr, dipped, needle, p.chunk, p.overlap, 0.13, p.min_distance, 2.0)
    assert [e[0] for e in exp] == offs
    algo = gpu.HipConvolve(needle)
    buf = gpu.DeviceBuffer.from_numpy(0, dipped)
    clean = gpu.DeviceBuffer.from_numpy(0, hay)
    try:
        one = algo.match_device(buf.ptr, dipped.size, p)
        assert_same(one, exp)
        for arm in (-1, 0):
            gpu.set_option("debug_redo_arm_at", arm)
            got = algo.match_batch_device([buf.ptr, clean.ptr, buf.ptr], [dipped.size, hay.size, dipped.size], p)
            assert key(got[0]) == key(got[2]) == key(one), arm
            assert [q.start for q in got[1]] == offs
    finally:
        gpu.set_option("debug_redo_arm_at", -2)
    # (b) a NaN near the end: behind everything the main pass reads (block 1 ends at hop + 2^22)
    bad = hay.copy()
    at = hay.size - 1000
    assert at > hop + (1 << 22)
    bad[at] = np.nan
    exp = oracle.calc_chunks(sr, bad, needle, p.chunk, p.overlap, 0.13, p.min_distance, 2.0)
    assert [e[0] for e in exp] == offs[:2]          # the last window holds the NaN: its hit is gone, T + 3 s (previous window) stays
    alone = algo.match(bad, p)
    assert_same(alone, exp)
    badbuf = gpu.DeviceBuffer.from_numpy(0, bad)
    got = algo.match_batch_device([clean.ptr, badbuf.ptr, clean.ptr], [hay.size, bad.size, hay.size], p)
    assert key(got[1]) == key(alone) and [q.start for q in got[0]] == [q.start for q in got[2]] == offs
    algo.close()
    # (c) i16 stereo, half_pipeline = 2
    def stereo(x):
        q = np.clip(np.round(x * 12000.0), -32768, 32767).astype(np.int16)
        return np.ascontiguousarray(np.stack([q, q], axis=1))
    ha = gpu.HipConvolve.from_pcm16(stereo(needle))
    ha.set_option("half_pipeline", 2)
    frames = stereo(hay)
    fbuf = gpu.DeviceBuffer.from_numpy(0, frames)
    res = {}
    try:
        for on in (1, 0):
            gpu.set_option("tail_block", on)
            res[on] = ha.match_pcm16(frames, p)
            assert [q.start for q in res[on]] == offs
            both = ha.match_pcm16_batch_device([fbuf.ptr, fbuf.ptr], [hay.size, hay.size], p)
            assert key(both[0]) == key(both[1]) == key(res[on]), on
        assert_close_peaks(res[1], res[0], tol=2e-3)
    finally:
        gpu.set_option("tail_block", 1)
        ha.close()


def test_odd_last_block_with_my_scaling_and_window_policy(gpu, oracle):
    """The tail block under the two rules that treat the END of a haystack specially: MyConvolve's scaling (the
    shorter windows at the end get their own factor and a pass of their own, audio_matcher.rs:442-448) and the
    policy switch tail_window = 1 (full-length windows only).  Haystack of 2 blocks of the 2^22 plan and a part-filled
    third, hits in the main pass, in the tail block's stretch and in the last, shorter window; offsets == checker,
    heights relative 1e-4, alone and in a batch, with the option on and off."""
    sr = 44100
    s = 10 * sr
    hop, hop_t = tail_geometry(s)
    T = 2 * hop
    chunk, overlap = 60 * sr, 10 * sr
    window = chunk + overlap
    needle, hay, offs, _ = tail_case(oracle, sr, s, 1900000, 221, lambda T, ht, out: [25 * sr, T + 2 * sr, out - 1 - 4 * sr])
    n_windows = -(-hay.size // chunk)
    assert (n_windows - 1) * chunk + window > hay.size and offs[2] > (n_windows - 1) * chunk    # the last window is a short one and holds a hit
    prom = 0.3 / window
    pm = gpu.AmMatchParams(sr=sr, chunk=chunk, overlap=overlap, min_prominence=prom, min_distance=2 * sr,
                           overshadow_distance_s=2.0, scale=int(gpu.Scale.MY))
    exp_my = oracle.calc_chunks(sr, hay, needle, chunk, overlap, prom, 2 * sr, 2.0, scale=oracle.SCALE_MY)
    assert [e[0] for e in exp_my] == offs
    pl = gpu.AmMatchParams(sr=sr, chunk=chunk, overlap=overlap, min_prominence=0.3, min_distance=2 * sr,
                           overshadow_distance_s=2.0, scale=int(gpu.Scale.LIB))
    exp_full = oracle.calc_chunks(sr, hay, needle, chunk, overlap, 0.3, 2 * sr, 2.0, pol=oracle.policy(tail_window=1))
    assert [e[0] for e in exp_full] == offs[:2]                     # the short last window is not emitted: its hit is gone
    algo = gpu.HipConvolve(needle)
    buf = gpu.DeviceBuffer.from_numpy(0, hay)
    rel = lambda got, exp: all(abs(g.height - e[2]) < 1e-4 * abs(e[2]) and abs(g.prominence - e[3]) < 1e-4 * abs(e[3]) for g, e in zip(got, exp))
    try:
        for on in (1, 0):
            gpu.set_option("tail_block", on)
            one = algo.match_device(buf.ptr, hay.size, pm)
            assert [(g.start, g.end) for g in one] == [(e[0], e[1]) for e in exp_my] and rel(one, exp_my), on
            both = algo.match_batch_device([buf.ptr, buf.ptr, buf.ptr], [hay.size] * 3, pm)
            assert all(key(b) == key(one) for b in both), on
            gpu.set_option("tail_window", 1)
            try:
                one = algo.match_device(buf.ptr, hay.size, pl)
                assert [(g.start, g.end) for g in one] == [(e[0], e[1]) for e in exp_full] and rel(one, exp_full), on
                both = algo.match_batch_device([buf.ptr, buf.ptr], [hay.size] * 2, pl)
                assert all(key(b) == key(one) for b in both), on
            finally:
                gpu.set_option("tail_window", 0)
    finally:
        gpu.set_option("tail_block", 1)
        gpu.set_option("tail_window", 0)
        algo.close()


def test_odd_last_block_with_several_needles(gpu, oracle):
    """The several-needle engine with the tail block: K1 of the tail once per haystack, its row kernel and K3 once per
    needle group behind the group's own launches, the block's ballots and thresholds preset for the group's needles.
    Three and nine needles of 10 s (groups of 8: nine = a group of eight and one single needle, for which the engine
    keeps the plain layout) against two haystacks with an odd last block and one without; every needle planted in the
    main pass, right in front of the boundary T and inside the tail; offsets == the checker == single calls, heights
    within 1e-4 -- with the option on and off, and with a NaN that only the tail's transform reads."""
    sr = 44100
    s = 10 * sr
    hop, hop_t = tail_geometry(s)
    T = 2 * hop
    p = gpu.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=2.0, prominence=0.3).params(sr, gpu.Scale.LIB)
    nn = 9
    needles = [oracle.synth_uniform(231, 10 + j, 0, s) for j in range(nn)]
    out = T + 1900000
    hays, plants = [], []
    for k, n in enumerate((out + s - 1, out + s - 1, 4 * hop - 5000 + s - 1)):        # two with a tail, one with four full blocks
        h = oracle.synth_uniform(231, 1 + k, 0, n)
        pk = []
        for j in range(nn):
            offs = sorted({(5 + 13 * j) * sr + 3 * k, T - 5 - (3 * j + 1) * 3 * sr if j % 2 else T + 7 + 3 * j * sr, T + hop_t - 2 + (j - 4) * 3 * sr})
            offs = [o for o in offs if o + s <= n]
            for o in offs:
                h[o:o + s] += needles[j]
            pk.append(offs)
        hays.append(h); plants.append(pk)
    algos = [gpu.HipConvolve(n_) for n_ in needles]
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
    exps = [[oracle.calc_chunks(sr, hays[0], n_, p.chunk, p.overlap, 0.3, p.min_distance, 2.0) for n_ in needles]]   # (the checker on the first haystack; the others through the single calls)
    assert [[e[0] for e in ex] for ex in exps[0]] == plants[0]
    try:
        for on in (1, 0):
            gpu.set_option("tail_block", on)
            singles = [[a.match_device(b.ptr, h.size, p) for a in algos] for b, h in zip(bufs, hays)]
            for count in (3, nn):
                res = gpu.match_multi_batch_device(algos[:count], [b.ptr for b in bufs], [h.size for h in hays], p)
                for k in range(3):
                    for j in range(count):
                        assert_close_peaks(res[k][j], singles[k][j])
                        if k == 0:
                            assert_same(res[k][j], exps[k][j])
                        else:
                            assert [q.start for q in res[k][j]] == plants[k][j]
            one = gpu.match_multi_device(algos[:3], bufs[0].ptr, hays[0].size, p)
            for j in range(3):
                assert_same(one[j], exps[0][j])
        # a NaN behind everything the main pass reads: the pair goes through the single-needle path, window by window
        gpu.set_option("tail_block", 1)
        bad = hays[0].copy()
        bad[bad.size - 1000] = np.nan
        bb = gpu.DeviceBuffer.from_numpy(0, bad)
        res = gpu.match_multi_device(algos[:3], bb.ptr, bad.size, p)
        for j in range(3):
            assert_same(res[j], oracle.calc_chunks(sr, bad, needles[j], p.chunk, p.overlap, 0.3, p.min_distance, 2.0))
    finally:
        gpu.set_option("tail_block", 1)
        for a in algos:
            a.close()
