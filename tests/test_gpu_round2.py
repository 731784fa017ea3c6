"""GPU tests of the round-2 boundary work, all through the C ABI:
the reference's overshadow / peak known answers driven through am_match itself, the
entry points no test called before, per-chunk progress, MyConvolve scaling in
calc_chunks, per-handle options, the dense / sparse / redo score paths and the
multi-device pool (am_pool_*)."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def key(r):
    return [(q.start, q.end, q.height, q.prominence) for q in r]


def assert_same(got, exp, tol=TOL):
    assert [g.start for g in got] == [e[0] for e in exp]
    assert [g.end for g in got] == [e[1] for e in exp]
    for g, e in zip(got, exp):
        assert abs(g.height - e[2]) < tol and abs(g.prominence - e[3]) < tol


# ---------------------------------------------------------------------------
# the reference's own known answers, through the shipped entry points
# ---------------------------------------------------------------------------
K2_DATA = [0.0, 0.7, 0.5, 1.0, 0.5, 0.8, 0.0]          # audio_matcher.rs:168


def spike_params(gpu, distance_s, min_distance=0):
    # sr = 1 as in the reference's test (audio_matcher.rs:189): sample index == seconds
    return gpu.AmMatchParams(sr=1, chunk=len(K2_DATA), overlap=0, min_prominence=0.0,
                             min_distance=min_distance, overshadow_distance_s=float(distance_s),
                             scale=int(gpu.Scale.NONE))


def test_k2_k3_known_answers_through_am_match(gpu):
    """audio_matcher.rs:167-218 through am_match: a one-sample needle [1.0] makes the Valid
    correlation the haystack itself, so find_peaks sees exactly the reference's test data.
    K2: prominences 0.2 / 1.0 / 0.3 at 1 / 3 / 5.  K3: with distance 3 s the neighbours of
    the tallest peak (2 s away, lower prominence) are overshadowed, with 2 s (strict <)
    nobody is; the tallest is never overshadowed."""
    algo = gpu.HipConvolve([1.0])
    hay = np.array(K2_DATA, np.float32)
    for _ in range(2):
        got = algo.match(hay, spike_params(gpu, 2))
        assert [(g.start, g.end) for g in got] == [(1, 2), (3, 4), (5, 6)]
        for g, prom, height in zip(got, (0.2, 1.0, 0.3), (0.7, 1.0, 0.8)):
            assert abs(g.prominence - prom) < 1e-6 and abs(g.height - height) < 1e-6   # reference tolerance (:181)
        got = algo.match(hay, spike_params(gpu, 3))
        assert [(g.start, g.prominence) for g in got] == [(3, 1.0)]
    # the same table through the device-resident and the batch entry points
    buf = gpu.DeviceBuffer.from_numpy(0, hay)
    assert [g.start for g in algo.match_device(buf.ptr, hay.size, spike_params(gpu, 3))] == [3]
    res = algo.match_batch_device([buf.ptr, buf.ptr], [hay.size, hay.size], spike_params(gpu, 2))
    assert [[g.start for g in r] for r in res] == [[1, 3, 5], [1, 3, 5]]


def test_k3_truth_table_pairs_through_am_match(gpu):
    """audio_matcher.rs:187-218 pair by pair: two peaks 2 s apart (sr = 1); the lower one is
    dropped iff distance > 2 s; equal prominence drops nobody (strict >)."""
    algo = gpu.HipConvolve([1.0])
    lo_hi = np.array([0, 0.2, 0, 1.0, 0], np.float32)      # peaks at 1 (prom .2) and 3 (prom 1)
    hi_lo = np.array([0, 1.0, 0, 0.3, 0], np.float32)      # peaks at 1 (prom 1) and 3 (prom .3)
    same = np.array([0, 0.5, 0, 0.5, 0], np.float32)
    for hay, tall in ((lo_hi, 3), (hi_lo, 1)):
        p = spike_params(gpu, 3); p.chunk = hay.size
        assert [g.start for g in algo.match(hay, p)] == [tall]
        p = spike_params(gpu, 2); p.chunk = hay.size
        assert [g.start for g in algo.match(hay, p)] == [1, 3]
    p = spike_params(gpu, 3); p.chunk = same.size
    assert [g.start for g in algo.match(same, p)] == [1, 3]


# measured on MI355X (profiles/r03/k1_kat.txt); the asserted bounds leave a factor of about two
K1_KAT_BOUNDS = {"direct": 1.2e-5, 10: 1.2e-5, 21: 1.2e-5, 22: 1.2e-5, 23: 1.2e-5}
# what the kernels measure today: two units in the last place of 52 (2 x 3.81e-6).  A change that only moves
# instructions around has cost a third ulp before (round 3, gpurun_out/r03x: 1.53e-5, over the reference's
# bound) -- this bound makes such a change visible one ulp BEFORE it reaches 1.2e-5.
K1_KAT_REGRESSION = {"direct": 0.0, 10: 7.63e-6, 21: 7.63e-6, 22: 7.63e-6, 23: 7.63e-6}
_k1_kat_cache = {}


def k1_kat_errors(gpu):
    """max |error| of the reference's correlation known answer (audio_matcher.rs:490-517) on the direct path and
    with every transform plan forced; also with the work matrix stored in f16 (recorded, not asserted: f16
    storage cannot hold 52 to 1.2e-5).  Written to gpurun_out/k1_kat.txt."""
    import os
    if _k1_kat_cache:
        return _k1_kat_cache["errs"], _k1_kat_cache["text"]
    within = np.arange(-10, 10, dtype=np.float32)
    expect = np.array([6 * j - 52 for j in range(18)], dtype=np.float32)
    algo = gpu.HipConvolve([1.0, 2.0, 3.0])
    errs = {"direct": float(np.abs(algo.correlate_with_sample(within, gpu.Mode.Valid, False) - expect).max())}
    for log_n in (10, 21, 22, 23):
        algo.set_option("log_n", log_n)
        errs[log_n] = float(np.abs(algo.correlate_with_sample(within, gpu.Mode.Valid, False) - expect).max())
    half = {}
    for log_n in (21, 22):
        algo.set_option("log_n", log_n)
        algo.set_option("half_pipeline", 1)
        half[log_n] = float(np.abs(algo.correlate_with_sample(within, gpu.Mode.Valid, False) - expect).max())
        algo.set_option("half_pipeline", -1)
    text = "K1 KAT (audio_matcher.rs:490-517), max |error| on values up to 52, reference bound 1.2e-5:\n" + \
           "".join(f"  {'direct summation' if k == 'direct' else 'transform 2^%d' % k}: {v:.3e} (asserted < {K1_KAT_BOUNDS[k]:.1e}, "
                   f"regression bound {K1_KAT_REGRESSION[k]:.2e})\n" for k, v in errs.items()) + \
           "".join(f"  transform 2^{k}, half_pipeline = 1 (f16 work matrix; recorded only): {v:.3e}\n" for k, v in half.items())
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "k1_kat.txt"), "w") as f:
        f.write(text)
    _k1_kat_cache["errs"], _k1_kat_cache["text"] = errs, text
    return errs, text


def test_k1_known_answer_at_reference_tolerance(gpu):
    """audio_matcher.rs:490-517 at the reference's own ABSOLUTE bound (|diff| < 1.2e-5, :511, on
    values up to 52) -- on the path the library picks for this input (direct summation: exact on
    integer data) and, with the plan forced, on the transform kernels themselves: the generic 2^10
    plan and the three register plans (2^21 = 256 x 8192, 2^22 = 512 x 8192, 2^23 = 1024 x 8192; the 20
    samples zero-padded into one block).  The measured errors go to gpurun_out/k1_kat.txt and into the
    failure text."""
    errs, text = k1_kat_errors(gpu)
    for k, v in errs.items():
        assert v < K1_KAT_BOUNDS[k], text


def test_k1_known_answer_regression_margin(gpu):
    """A REGRESSION bound, not the reference's: every transform plan reproduces the known answer to two
    units in the last place of 52 today (7.63e-6); the reference allows 3.1.  A rescheduling that costs the
    third ulp fails here first, while test_k1_known_answer_at_reference_tolerance still passes."""
    errs, text = k1_kat_errors(gpu)
    for k, v in errs.items():
        assert v <= K1_KAT_REGRESSION[k] * (1 + 1e-3), text


# ---------------------------------------------------------------------------
# entry points no test called before
# ---------------------------------------------------------------------------
def pcm_case(oracle, sr, needle_s, hay_s, plants_s, seed):
    rng = np.random.default_rng(seed)
    s, h = int(needle_s * sr), int(hay_s * sr)
    needle_lr = rng.integers(-9000, 9000, size=2 * s).astype(np.int16)
    hay_lr = rng.integers(-9000, 9000, size=2 * h).astype(np.int32)
    for t in plants_s:
        off = int(t * sr)
        hay_lr[2 * off:2 * (off + s)] += needle_lr
    return needle_lr, np.clip(hay_lr, -32768, 32767).astype(np.int16)


def test_pcm16_device_and_batch_entry_points(gpu, oracle):
    """am_match_pcm16_device / am_match_pcm16_batch_device / am_pcm_s16_stereo_to_mono_device
    (mp3_reader.rs:12, 28-37 down-mix, fused or as its own kernel) against the oracle."""
    sr = 48000
    needle_lr, hay_a = pcm_case(oracle, sr, 2.0, 75.0, [7.0, 51.25], 5)
    _, hay_b = pcm_case(oracle, sr, 2.0, 31.0, [], 6)
    off = int(19.5 * sr)
    hay_b = hay_b.astype(np.int32)
    hay_b[2 * off:2 * off + needle_lr.size] += needle_lr
    hay_b = np.clip(hay_b, -32768, 32767).astype(np.int16)
    needle = oracle.pcm_s16_stereo_to_mono(needle_lr)
    cfg = gpu.Config(chunk_size_s=30.0, overlap_length_s=2.0, distance_s=10.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve.from_pcm16(needle_lr)
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in (hay_a, hay_b)]
    frames = [hay_a.size // 2, hay_b.size // 2]
    exps = [oracle.calc_chunks(sr, oracle.pcm_s16_stereo_to_mono(h), needle, p.chunk, p.overlap, 0.13,
                               p.min_distance, 10.0) for h in (hay_a, hay_b)]
    assert [e[0] for e in exps[0]] == [7 * sr, int(51.25 * sr)] and [e[0] for e in exps[1]] == [off]
    for _ in range(2):                                  # dense, then sparse score path
        for b, f, e in zip(bufs, frames, exps):
            assert_same(algo.match_pcm16_device(b.ptr, f, p), e)
        res = algo.match_pcm16_batch_device([b.ptr for b in bufs] + [bufs[0].ptr], frames + [frames[0]], p)
        for r, e in zip(res, exps + [exps[0]]):
            assert_same(r, e)
    # the stand-alone down-mix on device memory: bit-exact
    mono = gpu.DeviceBuffer(0, 4 * frames[0])
    gpu._check(gpu.lib().am_pcm_s16_stereo_to_mono_device(0, bufs[0].ptr, frames[0], mono.ptr))
    assert np.array_equal(mono.to_numpy(np.float32, frames[0]), oracle.pcm_s16_stereo_to_mono(hay_a))
    # ... and what the f32 entry point finds on that buffer equals the fused ingest
    a32 = gpu.HipConvolve(needle)
    assert key(a32.match_device(mono.ptr, frames[0], p)) == key(algo.match_pcm16_device(bufs[0].ptr, frames[0], p))


# ---------------------------------------------------------------------------
# per-chunk progress (audio_matcher.rs:116-117, 129)
# ---------------------------------------------------------------------------
def test_per_chunk_progress_events(gpu, oracle):
    sr = 8000
    needle = oracle.synth_uniform(3, 0, 0, sr)
    hay = oracle.synth_uniform(3, 1, 0, 30 * sr)           # 3 chunks of 10 s
    cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0, distance_s=5.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    chunk_ev, hay_ev = [], []
    gpu.set_chunk_progress_callback(lambda k, i, n, stage: chunk_ev.append((k, i, n, stage)))
    gpu.set_progress_callback(lambda k, stage, n: hay_ev.append((k, stage, n)))
    try:
        algo.match(hay, p)
        assert chunk_ev == [(0, 0, 3, 0), (0, 1, 3, 0), (0, 2, 3, 0), (0, 0, 3, 1), (0, 1, 3, 1), (0, 2, 3, 1)]
        assert hay_ev == [(0, 0, 3), (0, 1, 3)]
        del chunk_ev[:], hay_ev[:]
        b = gpu.DeviceBuffer.from_numpy(0, hay)
        algo.match_batch_device([b.ptr, b.ptr], [hay.size, 12 * sr], p)    # 3 chunks and 2 chunks
        for k, n in ((0, 3), (1, 2)):
            mine = [e for e in chunk_ev if e[0] == k]
            assert mine == [(k, i, n, 0) for i in range(n)] + [(k, i, n, 1) for i in range(n)]
        assert hay_ev == [(0, 0, 3), (1, 0, 2), (0, 1, 3), (1, 1, 2)]
    finally:
        gpu.set_chunk_progress_callback(None)
        gpu.set_progress_callback(None)
    del chunk_ev[:]
    algo.match(hay, p)
    assert chunk_ev == []


# ---------------------------------------------------------------------------
# calc_chunks with MyConvolve's scaling (audio_matcher.rs:442-448)
# ---------------------------------------------------------------------------
def test_calc_chunks_with_my_scaling(gpu, oracle):
    """AM_SCALE_MY on am_match: every window is scaled by 1 / (sum(needle^2) * within.len()),
    so the shorter windows at the end of a haystack get their own factor (and prominences
    relative to it).  Relative tolerance: the scores are of order 1 / window."""
    sr = 8000
    s = 2 * sr
    needle = oracle.synth_uniform(7, 0, 0, s)
    hay = oracle.synth_uniform(7, 1, 0, 70 * sr)             # chunks at 0, 20, 40, 60 s; window 25 s
    for t in (5.0, 31.0, 47.5, 63.0):                        # the last plant sits in the 10 s tail window
        off = int(t * sr)
        hay[off:off + s] += needle
    window = 25 * sr
    prom = 0.3 / window                                      # a threshold in MyConvolve's units (0.12 for the 10 s tail window)
    p = gpu.AmMatchParams(sr=sr, chunk=20 * sr, overlap=5 * sr, min_prominence=prom, min_distance=0,
                          overshadow_distance_s=4.0, scale=int(gpu.Scale.MY))
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, prom, 0, 4.0, scale=oracle.SCALE_MY)
    assert [e[0] for e in exp] == [int(t * sr) for t in (5.0, 31.0, 47.5, 63.0)]
    assert exp[3][2] > 2 * exp[0][2]                          # the tail window's factor is larger
    algo = gpu.HipConvolve(needle)
    for _ in range(2):
        got = algo.match(hay, p)
        assert [(g.start, g.end) for g in got] == [(e[0], e[1]) for e in exp]
        for g, e in zip(got, exp):
            assert abs(g.height - e[2]) < 1e-4 * e[2] and abs(g.prominence - e[3]) < 1e-4 * e[3]
    # a haystack shorter than one window: only a "tail" window exists
    short = hay[: 12 * sr]
    exp = oracle.calc_chunks(sr, short, needle, p.chunk, p.overlap, 0.3 / (12 * sr), 0, 4.0, scale=oracle.SCALE_MY)
    p.min_prominence = 0.3 / (12 * sr)
    got = algo.match(short, p)
    assert [g.start for g in got] == [e[0] for e in exp] == [5 * sr]
    assert abs(got[0].height - exp[0][2]) < 1e-4 * exp[0][2]


# ---------------------------------------------------------------------------
# options: per call / per handle
# ---------------------------------------------------------------------------
def test_per_handle_plan_and_concurrent_calls(gpu, oracle):
    """log_n is a property of the handle (am_needle_set_option), not of the process: two
    handles with different plans used from two threads at once give the answers they give
    alone, and the process default is untouched."""
    sr = 8000
    needle = oracle.synth_uniform(9, 0, 0, 3 * sr)
    hay = oracle.synth_uniform(9, 1, 0, 120 * sr)
    for t in (10.0, 77.0):
        off = int(t * sr)
        hay[off:off + needle.size] += needle
    cfg = gpu.Config(chunk_size_s=30.0, overlap_length_s=3.0, distance_s=10.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 10.0)
    a, b = gpu.HipConvolve(needle), gpu.HipConvolve(needle)
    a.set_option("log_n", 16)
    b.set_option("log_n", 18)
    assert (a.get_option("log_n"), b.get_option("log_n"), gpu.get_option("log_n")) == (16, 18, 0)
    alone = [key(a.match(hay, p)), key(b.match(hay, p))]
    out = [[], []]

    def work(i, algo):
        for _ in range(6):
            out[i].append(key(algo.match(hay, p)))
    th = [threading.Thread(target=work, args=(0, a)), threading.Thread(target=work, args=(1, b))]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in (0, 1):
        assert all(r == alone[i] for r in out[i])
    assert_same(a.match(hay, p), exp)
    assert_same(b.match(hay, p), exp)
    with pytest.raises(gpu.AudioMatchError):
        a.set_option("log_n", 99)
    a.set_option("log_n", -1)
    assert a.get_option("log_n") == -1


def test_dense_scores_option_is_result_neutral(gpu, oracle):
    """dense_scores = 1 (every raw score written, the worst case of the sparse path) gives
    bit-identical results to the sparse path."""
    sr = 44100
    needle = oracle.synth_uniform(13, 0, 0, 4 * sr)
    hay = oracle.synth_uniform(13, 1, 0, 200 * sr)
    for t in (33.0, 150.5):
        off = int(t * sr)
        hay[off:off + needle.size] += needle
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=4.0, distance_s=30.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    first = key(algo.match(hay, p))            # sparse, no history yet
    sparse = key(algo.match(hay, p))           # sparse, threshold bounded by the first call's minima
    gpu.set_option("dense_scores", 1)
    try:
        dense = key(algo.match(hay, p))
    finally:
        gpu.set_option("dense_scores", 0)
    assert first == sparse == dense and [q[0] for q in first] == [33 * sr, int(150.5 * sr)]


def test_failed_certificate_redoes_only_that_chunk(gpu, oracle):
    """A chunk whose minimum lies far below what the K3 tiles sampled (an inverted copy of the
    needle: a dip to -1 a few scores wide, missed by all but one of the 256 tiles) fails its
    certificate; the chunk is redone in place with every run written and the result equals the
    oracle and a fresh handle's answer bit for bit, also inside a batch."""
    sr = 44100
    s = 3 * sr
    needle = oracle.synth_uniform(15, 0, 0, s)
    calm = oracle.synth_uniform(15, 1, 0, 200 * sr)
    calm[20 * sr:20 * sr + s] += needle
    wild = oracle.synth_uniform(15, 2, 0, 200 * sr)
    wild[70 * sr:70 * sr + s] -= needle                      # chunk 1: minimum ~ -1
    wild[75 * sr:75 * sr + s] += needle                      # ... and a hit in the same chunk
    wild[150 * sr:150 * sr + s] += needle
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=3.0, distance_s=2.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, wild, needle, p.chunk, p.overlap, 0.13, p.min_distance, 2.0)
    assert [e[0] for e in exp] == [75 * sr, 150 * sr]
    fresh = key(gpu.HipConvolve(needle).match(wild, p))
    algo = gpu.HipConvolve(needle)
    algo.match(calm, p)                                      # recent minimum ~ -0.03
    got = algo.match(wild, p)                                # chunk 1 fails its certificate
    assert key(got) == fresh
    assert_same(got, exp)
    # the same inside a batch (score buffers of the redone haystack were reused meanwhile)
    algo2 = gpu.HipConvolve(needle)
    algo2.match(calm, p)
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in (calm, wild, calm)]
    res = algo2.match_batch_device([b.ptr for b in bufs], [calm.size] * 3, p)
    assert key(res[1]) == fresh and [q.start for q in res[0]] == [20 * sr] == [q.start for q in res[2]]


def test_more_peaks_than_a_header_holds_in_a_batch(gpu, oracle):
    """Chunks with more than four peaks spill their list to the result arena instead of
    forcing a second pass; in a batch every haystack still equals its single call."""
    sr = 8000
    needle = oracle.synth_uniform(19, 0, 0, sr // 2)
    hays = []
    for k in range(3):
        h = oracle.synth_uniform(19, 1 + k, 0, 45 * sr)
        for t in np.arange(1.0 + k, 44.0, 2.5):              # ~6 plants per 15 s chunk
            off = int(t * sr)
            h[off:off + needle.size] += needle
        hays.append(h)
    cfg = gpu.Config(chunk_size_s=15.0, overlap_length_s=0.5, distance_s=1.0, prominence=0.4)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    bufs = [gpu.DeviceBuffer.from_numpy(0, h) for h in hays]
    for _ in range(2):
        res = algo.match_batch_device([b.ptr for b in bufs], [h.size for h in hays], p)
        for r, h in zip(res, hays):
            exp = oracle.calc_chunks(sr, h, needle, p.chunk, p.overlap, 0.4, p.min_distance, 1.0)
            assert len(exp) >= 15
            assert_same(r, exp)


# ---------------------------------------------------------------------------
# the pool: matcher::run's file loop sharded over devices (matcher/mod.rs:42-87)
# ---------------------------------------------------------------------------
def pool_inputs(oracle, sr):
    needle = oracle.synth_uniform(23, 0, 0, 2 * sr)
    secs = [40.0, 1.0, 95.0, 33.3, 61.0, 0.0, 12.0]
    hays = []
    for k, t in enumerate(secs):
        h = oracle.synth_uniform(23, 10 + k, 0, int(t * sr))
        if t > 10:
            off = int(0.4 * t * sr) + 17 * k
            h[off:off + needle.size] += needle
        hays.append(h)
    return needle, hays


def test_pool_equals_single_device_calls(gpu, oracle):
    sr = 22050
    needle, hays = pool_inputs(oracle, sr)
    cfg = gpu.Config(chunk_size_s=30.0, overlap_length_s=2.0, distance_s=10.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    singles = [key(algo.match(h, p)) if h.size else [] for h in hays]
    assert [len(r) for r in singles] == [1, 0, 1, 1, 1, 0, 1]
    for devices in ([0], [0, 0], [0, 0, 0]):              # several slots on one device share its queue
        pool = gpu.Pool(needle, devices)
        assert pool.size == len(devices) and pool.device_of(len(devices) - 1) == 0
        for _ in range(2):
            res = pool.match_batch(hays, p)               # host buffers: ring + copier thread per slot
            assert [key(r) for r in res] == singles, devices
        bufs = [gpu.DeviceBuffer.from_numpy(0, h) if h.size else None for h in hays]
        res = pool.match_batch_device([b.ptr if b else None for b in bufs], [h.size for h in hays], p)
        assert [key(r) for r in res] == singles, devices
        assert pool.match_batch([], p) == []
        pool.close()
    every = gpu.Pool(needle)                               # devices = NULL: all visible devices
    assert every.size == gpu.device_count()
    assert [key(r) for r in every.match_batch(hays, p)] == singles


def test_pool_progress_uses_batch_indices_and_capacity(gpu, oracle):
    sr = 22050
    needle, hays = pool_inputs(oracle, sr)
    cfg = gpu.Config(chunk_size_s=30.0, overlap_length_s=2.0, distance_s=10.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    pool = gpu.Pool(needle, [0, 0])
    ev = []
    lock = threading.Lock()

    def on(k, stage, n):
        with lock:
            ev.append((k, stage, n))
    gpu.set_progress_callback(on)
    try:
        pool.match_batch(hays, p)
    finally:
        gpu.set_progress_callback(None)
    active = [k for k, h in enumerate(hays) if h.size >= needle.size]
    assert sorted(e[0] for e in ev if e[1] == 0) == active and sorted(e[0] for e in ev if e[1] == 1) == active
    # a too small output capacity is reported, with the counts filled in
    k = len(hays)
    arr_p = (C.c_void_p * k)(*[h.ctypes.data if h.size else None for h in hays])
    arr_l = (C.c_size_t * k)(*[h.size for h in hays])
    counts = (C.c_size_t * k)()
    rc = gpu.lib().am_pool_match_batch(pool._p, arr_p, arr_l, k, C.byref(p), None, 0, counts)
    assert rc == gpu.AM_ERR_CAPACITY and list(counts) == [1, 0, 1, 1, 1, 0, 1]


# ---------------------------------------------------------------------------
# score arrays that are not white: thousands of candidate maxima per chunk
# ---------------------------------------------------------------------------
def drifting_case(oracle, sr, seed):
    """Needle with a DC offset and a 50 Hz tone, haystack with the same tone and a slow drift:
    the scores drift by about +-0.13 (monotone inside a chunk) with a +-0.03 ripple on top."""
    rng = np.random.default_rng(seed)
    s, h = 2 * sr, 200 * sr
    t = np.arange(h, dtype=np.float64)
    tone = (0.0437 * np.sin(2 * np.pi * 50.0 / sr * t)).astype(np.float32)
    drift = (0.04 * np.sin(2 * np.pi * t / (80.0 * sr))).astype(np.float32)
    needle = rng.uniform(-0.25, 0.25, s).astype(np.float32) + tone[:s] + np.float32(0.1)
    hay = rng.uniform(-0.25, 0.25, h).astype(np.float32) + tone + drift
    plants = [int(13.3 * sr), int(95.0 * sr) + 7, int(171.2 * sr)]
    for p0 in plants:
        hay[p0:p0 + s] += needle
    return needle, hay, plants


def test_many_candidate_maxima_default_distance(gpu, oracle):
    """Chunks with hundreds of candidate tiles (the multi-workgroup peak pick) and the
    reference's default regime min_distance >= chunk: equal to the oracle, on the dense and the
    sparse score path; only the planted hits qualify."""
    sr = 8000
    needle, hay, plants = drifting_case(oracle, sr, 31)
    cfg = gpu.Config(chunk_size_s=20.0, overlap_length_s=2.0, distance_s=480.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 480.0)
    assert [e[0] for e in exp] == [plants[0]]            # 480 s apart: the tallest overshadows the other two
    cfg2 = gpu.Config(chunk_size_s=20.0, overlap_length_s=2.0, distance_s=30.0, prominence=0.13)
    p2 = cfg2.params(sr, gpu.Scale.LIB)
    exp2 = oracle.calc_chunks(sr, hay, needle, p2.chunk, p2.overlap, 0.13, p2.min_distance, 30.0)
    assert [e[0] for e in exp2] == plants
    algo = gpu.HipConvolve(needle)
    for _ in range(2):
        assert_same(algo.match(hay, p), exp)
        assert_same(algo.match(hay, p2), exp2)


def test_many_qualifying_maxima_small_distance(gpu, oracle):
    """A low prominence threshold lets every ripple maximum qualify (hundreds per chunk) and
    a short min_distance keeps dozens of them: the window test accepts most without a walk,
    the chunk's pieces are spread over several workgroups, the list is sorted and filtered
    afterwards -- all of it must equal the oracle peak for peak."""
    sr = 8000
    needle, hay, plants = drifting_case(oracle, sr, 37)
    for chunk_s, dist_s, prom in ((10.0, 1.0, 0.04), (10.0, 0.0, 0.05), (15.0, 3.0, 0.045)):
        cfg = gpu.Config(chunk_size_s=chunk_s, overlap_length_s=2.0, distance_s=dist_s, prominence=prom)
        p = cfg.params(sr, gpu.Scale.LIB)
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, prom, p.min_distance, dist_s, cap=1 << 16)
        assert len(exp) > 30
        algo = gpu.HipConvolve(needle)
        for _ in range(2):
            got = algo.match(hay, p, cap=1 << 16)
            assert_same(got, exp)


def test_more_than_1024_qualifying_maxima_per_chunk(gpu, oracle):
    """More peaks pass the prominence filter in one chunk than a workgroup can list in LDS
    (AM_MAX_PEAKS_PER_CHUNK = 1024): find_peaks returns them all (audio_matcher.rs:221-230), so the
    chunk goes through the global-memory list, the radix sort by (height, position) and the
    bucketed greedy distance filter -- with no distance (every one of them comes back), a short one
    and a long one; single calls (dense, then sparse scores), a batch of two (the score buffers have
    moved on when the overflow is noticed: the chunk's blocks are redone) and MyConvolve scaling
    with its separately correlated tail windows."""
    sr = 8000
    needle, hay, plants = drifting_case(oracle, sr, 37)
    algo = gpu.HipConvolve(needle)
    # With nearly twenty thousand peaks some prominence always lies within rounding of any bound
    # (0.02 itself has one 4e-8 above it).  Take a bound that the checker's values clear by 2e-6 on
    # both sides, so that the comparison does not hinge on the last bit of a score.
    c40 = gpu.Config(chunk_size_s=40.0, overlap_length_s=2.0, distance_s=0.0, prominence=0.02).params(sr, gpu.Scale.LIB)
    thr = None
    for k in range(60):
        cand = 0.02 + 1e-4 * k
        near = oracle.calc_chunks(sr, hay, needle, c40.chunk, c40.overlap, cand - 2e-6, 0, 0.0, cap=1 << 20)
        if not any(abs(e[3] - cand) <= 2e-6 for e in near):
            thr = cand
            break
    assert thr is not None
    for chunk_s, dist_s, prom in ((40.0, 0.0, thr), (40.0, 0.01, thr), (40.0, 2.5, thr)):
        cfg = gpu.Config(chunk_size_s=chunk_s, overlap_length_s=2.0, distance_s=dist_s, prominence=prom)
        p = cfg.params(sr, gpu.Scale.LIB)
        per_chunk = oracle.find_peaks(oracle.correlate(hay[:p.chunk + p.overlap], needle, oracle.MODE_VALID, oracle.SCALE_LIB),
                                      prom, 0, cap=1 << 20)
        assert len(per_chunk) > 1100, len(per_chunk)
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, prom, p.min_distance, dist_s, cap=1 << 20)
        assert len(exp) > 30
        for _ in range(2):
            assert_same(algo.match(hay, p, cap=1 << 20), exp)
    bufs = [gpu.DeviceBuffer.from_numpy(0, hay), gpu.DeviceBuffer.from_numpy(0, hay[: hay.size // 2])]
    exp2 = oracle.calc_chunks(sr, hay[: hay.size // 2], needle, p.chunk, p.overlap, prom, p.min_distance, dist_s, cap=1 << 20)
    res = algo.match_batch_device([b.ptr for b in bufs], [hay.size, hay.size // 2], p, cap_per_hay=1 << 16)
    assert_same(res[0], exp)
    assert_same(res[1], exp2)
    pm = cfg.params(sr, gpu.Scale.MY)
    expm = oracle.calc_chunks(sr, hay, needle, pm.chunk, pm.overlap, prom / (pm.chunk + pm.overlap), pm.min_distance, dist_s,
                              cap=1 << 20, scale=oracle.SCALE_MY)
    pm.min_prominence = prom / (pm.chunk + pm.overlap)
    assert_same(algo.match(hay, pm, cap=1 << 20), expm, tol=1e-8)


def test_default_distance_with_thousands_of_qualifying_maxima(gpu, oracle):
    """min_distance >= chunk (the reference's default regime) on a score array whose ripple is
    larger than min_prominence: about two thousand maxima per chunk pass the prominence filter --
    more than any list holds -- and the distance filter keeps the tallest.  With a rising drift
    the chunk's maximum sits at the chunk's end and fails the prominence test itself, so the
    answer has to come from the general path (a running maximum, nothing to overflow)."""
    sr = 8000
    rng = np.random.default_rng(43)
    s, h = 2 * sr, 200 * sr
    t = np.arange(h, dtype=np.float64)
    tone = (0.0827 * np.sin(2 * np.pi * 50.0 / sr * t)).astype(np.float32)      # score ripple about +-0.1
    drift = (0.04 * np.sin(2 * np.pi * t / (160.0 * sr))).astype(np.float32)
    needle = rng.uniform(-0.25, 0.25, s).astype(np.float32) + tone[:s] + np.float32(0.1)
    hay = rng.uniform(-0.25, 0.25, h).astype(np.float32) + tone + drift
    hay[int(95.0 * sr) + 7:int(95.0 * sr) + 7 + s] += needle
    cfg = gpu.Config(chunk_size_s=40.0, overlap_length_s=2.0, distance_s=45.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 45.0)
    sc = oracle.correlate(hay[: 42 * sr], needle, oracle.MODE_VALID, oracle.SCALE_LIB)
    assert len(oracle.find_peaks(sc, 0.13, 0, cap=1 << 16)) > 1500        # far beyond AM_MAX_PEAKS_PER_CHUNK
    assert len(exp) >= 2 and int(95.0 * sr) + 7 in [e[0] for e in exp]
    algo = gpu.HipConvolve(needle)
    for _ in range(2):
        assert_same(algo.match(hay, p), exp)
    # The hard case made on purpose, through find_peaks on a given score array: a steep ramp under
    # the ripple, cut off 20 scores behind a crest.  The array's maximum is that last crest, whose
    # right side ends at the edge before it has dropped by min_prominence: it fails, about two
    # thousand others pass, and the answer is the tallest of those.
    for seed in range(3):
        n = 160 * 2000 + 60
        tt = np.arange(n, dtype=np.float64)
        y = (2e-5 * tt + 0.1 * np.sin(2 * np.pi * tt / 160.0) +
             0.002 * np.random.default_rng(seed).standard_normal(n)).astype(np.float32)
        ref = oracle.find_peaks(y, 0.13, n)
        assert len(oracle.find_peaks(y, 0.13, 0, cap=1 << 16)) > 1500 and len(ref) == 1
        assert ref[0][0] < n - 100                                          # not the last crest
        got = gpu.find_peaks(y, 0.13, n)
        assert [(g.start, g.end, g.height, g.prominence) for g in got] == [tuple(r) for r in ref]


def test_512_row_plan_on_pcm16_and_several_needles(gpu, oracle):
    """The N = 2^22 plan (512 x 8192, 512-thread column kernels; needles above 300 000 samples) on
    the two other ingest paths: interleaved i16 stereo frames (down-mix fused into K1) and several
    needles against one haystack (grouped K2 over 512 rows per pair)."""
    sr = 48000
    needle_lr, hay_lr = pcm_case(oracle, sr, 7.0, 95.0, [11.0, 70.5], 61)        # 336 000 frames per needle
    needle = oracle.pcm_s16_stereo_to_mono(needle_lr)
    hay = oracle.pcm_s16_stereo_to_mono(hay_lr)
    cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=7.0, distance_s=30.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 30.0)
    assert [e[0] for e in exp] == [11 * sr, int(70.5 * sr)]
    algo = gpu.HipConvolve.from_pcm16(needle_lr)
    for _ in range(2):
        assert_same(algo.match_pcm16(hay_lr, p), exp)
    # three needles of that length, one haystack: equal to separate calls
    others = [oracle.synth_uniform(61, 200 + k, 0, needle.size) for k in range(2)]
    hay2 = hay.copy()
    for k, n2 in enumerate(others):
        off = int((25.0 + 30.0 * k) * sr)
        hay2[off:off + n2.size] += n2
    algos = [gpu.HipConvolve(needle)] + [gpu.HipConvolve(n2) for n2 in others]
    buf = gpu.DeviceBuffer.from_numpy(0, hay2)
    res = gpu.match_multi_device(algos, buf.ptr, hay2.size, p)
    singles = [a.match_device(buf.ptr, hay2.size, p) for a in algos]
    assert [[q.start for q in r] for r in res] == [[11 * sr, int(70.5 * sr)], [25 * sr], [55 * sr]]
    for got, one in zip(res, singles):
        assert [(g.start, g.end) for g in got] == [(g.start, g.end) for g in one]
        for g, o in zip(got, one):
            assert abs(g.height - o.height) < 2e-6 and abs(g.prominence - o.prominence) < 2e-6


@pytest.mark.parametrize("level", [1, 2])
def test_half_pipeline_levels_on_both_plans_and_several_needles(gpu, oracle, level):
    """Option half_pipeline on the shapes test_half_pipeline_config5 (tests/test_gpu_match.py) does not
    reach: the N = 2^21 plan (5 s needle: level 2 there means K2's f16 butterflies between the f32
    column kernels) and several needles against one haystack (K2 into a second work matrix) on both
    plans; per handle, so that nothing leaks into other tests.  Offsets identical, scores to 1e-3."""
    sr = 44100
    for secs, seed in ((5.0, 71), (8.0, 72)):               # 220 500 samples: 2^21; 352 800 samples: 2^22
        s = int(secs * sr)
        needles = [oracle.synth_uniform(seed, 300 + k, 0, s) for k in range(2)]
        hay = oracle.synth_uniform(seed, 1, 0, 130 * sr)
        plants = [[int(12.25 * sr), int(91.0 * sr)], [int(47.5 * sr)]]
        for n_, offs in zip(needles, plants):
            for off in offs:
                hay[off:off + s] += n_
        cfg = gpu.Config(chunk_size_s=60.0, overlap_length_s=secs, distance_s=20.0, prominence=0.13)
        p = cfg.params(sr, gpu.Scale.LIB)
        algos = [gpu.HipConvolve(n_) for n_ in needles]
        for a in algos:
            a.set_option("half_pipeline", level)
        buf = gpu.DeviceBuffer.from_numpy(0, hay)
        exps = [oracle.calc_chunks(sr, hay, n_, p.chunk, p.overlap, 0.13, p.min_distance, 20.0) for n_ in needles]
        assert [[e[0] for e in ex] for ex in exps] == plants
        for _ in range(2):                                  # dense first call, then the sparse score path
            for a, ex in zip(algos, exps):
                assert_same(a.match_device(buf.ptr, hay.size, p), ex, tol=1e-3)
        res = gpu.match_multi_device(algos, buf.ptr, hay.size, p)   # (takes the first handle's options)
        for got, ex in zip(res, exps):
            assert_same(got, ex, tol=1e-3)


def test_non_finite_samples_cost_only_their_own_windows(gpu, oracle):
    """A NaN / infinity among the samples: the reference transforms every window on its own
    (audio_matcher.rs:114-122), so exactly the windows that hold the sample lose their peaks.  The
    overlap-save blocks here are longer than a window and a bad sample poisons the whole block pair:
    the library has to notice, drop the windows that hold the sample and correlate the clean windows
    of that pair again.  Checked against the checker's per-window result on the register plan
    (8 kHz x 100 s), the generic plan (30 s), a batch with one bad haystack, and the direct-summation
    path (48-sample needle)."""
    sr = 8000
    s = 2 * sr
    needle = oracle.synth_uniform(5, 0, 0, s)
    cfg = gpu.Config(chunk_size_s=10.0, overlap_length_s=2.0, distance_s=5.0, prominence=0.13)
    p = cfg.params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)

    def case(secs, plants, bad_at):
        hay = oracle.synth_uniform(5, 1, 0, secs * sr)
        for t in plants:
            hay[t * sr:t * sr + s] += needle
        bad = hay.copy()
        for j, v in zip(bad_at, (np.nan, np.inf, -np.inf)):
            bad[j] = v
        return hay, bad

    for secs, plants, bad_at in ((100, (13, 35, 57, 81), (34 * sr, 34 * sr + 5)),        # one window lost
                                 (100, (13, 35, 57, 81), (31 * sr, 31 * sr + 1, 31 * sr + 2)),  # in the overlap: two windows
                                 (30, (3, 14, 25), (12 * sr + 7,))):                    # generic plan
        hay, bad = case(secs, plants, bad_at)
        exp_clean = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
        exp_bad = oracle.calc_chunks(sr, bad, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
        assert len(exp_bad) < len(exp_clean) == len(plants)
        for _ in range(2):                                   # dense, then sparse score path
            assert_same(algo.match(bad, p), exp_bad)
        assert_same(algo.match(hay, p), exp_clean)           # and nothing sticks to the handle
    # level 1: one window, one transform in the reference -- every output is NaN
    win = case(30, (3, 14, 25), (12 * sr + 7,))[1]
    for mode in (gpu.Mode.Valid, gpu.Mode.Same, gpu.Mode.Full):
        assert np.isnan(algo.correlate_with_sample(win, mode, True)).all()
    assert np.isnan(oracle.correlate(win, needle, oracle.MODE_VALID, oracle.SCALE_LIB)).all()
    assert np.isfinite(algo.correlate_with_sample(win[: 12 * sr], gpu.Mode.Valid, True)).all()
    # a batch: only the bad haystack takes the slow path
    hay, bad = case(100, (13, 35, 57, 81), (34 * sr,))
    exp_clean = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
    exp_bad = oracle.calc_chunks(sr, bad, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0)
    bufs = [gpu.DeviceBuffer.from_numpy(0, x) for x in (hay, bad, hay)]
    res = algo.match_batch_device([b.ptr for b in bufs], [hay.size] * 3, p)
    for got, exp in zip(res, (exp_clean, exp_bad, exp_clean)):
        assert_same(got, exp)
    # direct summation (needle of at most 64 samples): a bad sample reaches only s scores, the
    # reference still loses the whole window
    tiny = oracle.synth_uniform(6, 0, 0, 48)
    th = oracle.synth_uniform(6, 1, 0, 40 * sr)
    for t in (5, 17, 33):
        th[t * sr:t * sr + 48] += 4.0 * tiny
    tb = th.copy(); tb[16 * sr] = np.nan
    pt = gpu.Config(chunk_size_s=10.0, overlap_length_s=1.0, distance_s=5.0, prominence=2.0).params(sr, gpu.Scale.LIB)
    exp_t = oracle.calc_chunks(sr, tb, tiny, pt.chunk, pt.overlap, 2.0, pt.min_distance, 5.0)
    assert [e[0] for e in exp_t] == [5 * sr, 33 * sr]
    assert_same(gpu.HipConvolve(tiny).match(tb, pt), exp_t)


def test_profile_counters_and_pool_slots(gpu, oracle):
    """am_profile_* (the HIP-event timing bench.py's roofline comes from) and am_pool_slot: counts
    and times appear for the kernels that ran, a name that does not exist is an error, reset clears,
    disabled costs nothing; a pool reports the device behind every slot."""
    sr = 8000
    needle = oracle.synth_uniform(9, 0, 0, 2 * sr)
    hay = oracle.synth_uniform(9, 1, 0, 100 * sr)           # 800 000 samples: the register kernels
    hay[43 * sr:45 * sr] += needle
    p = gpu.Config(chunk_size_s=10.0, overlap_length_s=2.0, distance_s=5.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    algo = gpu.HipConvolve(needle)
    algo.match(hay, p)
    gpu.set_option("profile_mask", -1)
    with gpu.Profile(0) as prof:
        for _ in range(3):
            assert [q.start for q in algo.match(hay, p)] == [43 * sr]
        k2_ms, k2_n = prof.query("k2_rows")
        all_ms, all_n = prof.query("*")
        assert k2_n == 3 and 0.0 < k2_ms < 50.0
        assert all_n >= 3 * 5 and all_ms >= k2_ms
        for name in ("k1_cols_fwd", "k3_cols_inv", "tile_stats", "peaks"):
            assert prof.query(name)[1] >= 3
        with pytest.raises(gpu.AudioMatchError):
            prof.query("no_such_kernel")
    with gpu.Profile(0) as prof:                             # entering resets
        assert prof.query("*")[1] == 0
    algo.match(hay, p)                                       # disabled again: nothing is recorded
    ms, n = C.c_double(0), C.c_uint64(0)
    assert gpu.lib().am_profile_query(0, b"*", C.byref(ms), C.byref(n)) == 0 and n.value == 0
    pool = gpu.Pool(needle, [0, 0, 0])
    assert pool.size == 3 and [pool.device_of(k) for k in range(3)] == [0, 0, 0]
    with pytest.raises(gpu.AudioMatchError):
        pool.device_of(3)
    pool.close()
