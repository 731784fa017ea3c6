"""GPU tests: the unpinned rules as options of the library (am_set_option "peak_filter_order", "distance_rule",
"tail_window", "surrounding_from"; defaults unchanged) give exactly what the checker gives under the same policy
(oracle.h orc_policy) -- on the designed inputs of tests/policy_cases.py, where the variants differ, through
am_find_peaks and through am_match (sparse scores, the > 1024-peaks path, the default-distance fast path)."""
import numpy as np
import pytest

import policy_cases as pc

pytestmark = pytest.mark.gpu

OPTS = ("peak_filter_order", "distance_rule", "tail_window", "surrounding_from")


def key(r):
    return [(q.start, q.end, q.height, q.prominence) for q in r]


class policy_set:
    def __init__(self, gpu, **kw):
        self.gpu, self.kw = gpu, kw

    def __enter__(self):
        for k, v in self.kw.items():
            self.gpu.set_option(k, v)

    def __exit__(self, *exc):
        for k in OPTS:
            self.gpu.set_option(k, 0)


@pytest.mark.parametrize("case", sorted(pc.PEAK_CASES))
def test_find_peaks_equals_checker_under_every_peak_policy(gpu, oracle, case):
    """am_find_peaks (find_peaks, audio_matcher.rs:221-230) bit for bit against the checker for both filter orders x
    the four distance rules; the variants really differ on these inputs."""
    make, prom, dist = pc.PEAK_CASES[case]
    x = make()
    seen = set()
    for order, rule in pc.PEAK_POLICIES:
        exp = oracle.find_peaks(x, prom, dist, pol=oracle.policy(order, rule))
        with policy_set(gpu, peak_filter_order=order, distance_rule=rule):
            got = gpu.find_peaks(x, prom, dist)
        assert key(got) == exp, (case, order, rule)
        seen.add(tuple(e[0] for e in exp))
    assert len(seen) >= 2, case
    assert gpu.get_option("peak_filter_order") == 0 and gpu.get_option("distance_rule") == 0


def test_many_maxima_take_the_big_list_path_under_the_distance_first_order(gpu, oracle):
    """distance -> prominence makes EVERY maximum that passes the height test a candidate of the distance filter: far
    more than the 1024 a workgroup orders on chip.  60 000 scores, ~9 000 maxima, min_distance 25."""
    rng = np.random.default_rng(21)
    x = (0.1 * rng.standard_normal(60000)).astype(np.float32)
    x[::997] += np.float32(0.8)
    for order in (0, 1):
        for rule in (0, 3):
            exp = oracle.find_peaks(x, 0.3, 25, pol=oracle.policy(order, rule))
            with policy_set(gpu, peak_filter_order=order, distance_rule=rule):
                got = gpu.find_peaks(x, 0.3, 25)
            assert key(got) == exp, (order, rule)


def test_match_equals_checker_under_every_policy(gpu, oracle):
    """am_match (calc_chunks, audio_matcher.rs:88-141) == the checker for the tail-window rule x the neighbour rule x
    both filter orders, on a haystack built to tell them apart; offsets identical, scores within 1e-4."""
    sr = 8000
    needle, hay, plants = pc.chunk_case(oracle.synth_uniform, sr)
    s = needle.size
    p = gpu.Config(chunk_size_s=30.0, overlap_length_s=1.0, distance_s=3.5, prominence=0.13).params(sr, gpu.Scale.LIB)
    p.min_distance = sr
    algo = gpu.HipConvolve(needle)
    seen = set()
    for tail in (0, 1):
        for surr in (0, 1):
            for order in (0, 1):
                exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 3.5,
                                         pol=oracle.policy(order, 0, tail, surr))
                with policy_set(gpu, peak_filter_order=order, tail_window=tail, surrounding_from=surr):
                    got = algo.match(hay, p)
                    plan = [gpu.long_plan(hay.size, s, p, 2, i) for i in range(2)]
                    pool = gpu.Pool(needle, [0, 0])
                    got_long = pool.match_long(hay, p)
                    pool.close()
                assert [g.start for g in got] == [e[0] for e in exp], (tail, surr, order)
                assert [g.start for g in got_long] == [e[0] for e in exp], (tail, surr, order)
                assert sum(pl[1] for pl in plan) == (3 if tail else 4)
                for g, e in zip(got, exp):
                    assert abs(g.height - e[2]) < 1e-4 and abs(g.prominence - e[3]) < 1e-4
                seen.add(tuple(e[0] for e in exp))
    assert len(seen) == 4           # (tail, neighbour) each change the answer; the filter order does not on this input


def test_default_regime_fast_path_under_the_distance_first_order(gpu, oracle):
    """The reference's default regime (min_distance >= chunk length) through am_match on a score array whose tallest
    maximum in a chunk is not prominent: needle and haystack share a DC component that rises slowly through chunk 1
    and on into chunk 2 (the score floor follows it, 0 -> 0.33 by the end of chunk 1), with a quiet hit (0.2) earlier in
    chunk 1.  prominence -> distance returns the hit at 40 s; distance -> prominence nothing for that chunk (the last
    noise crest of the rising floor is taller, suppresses the hit and is then dropped for its prominence of ~0.03) --
    both as the checker has it, on the first call of a handle as on the second."""
    sr = 8000
    s = sr
    needle = oracle.synth_uniform(43, 0, 0, s) + np.float32(0.1)
    hay = oracle.synth_uniform(43, 1, 0, 90 * sr)
    hay[10 * sr:10 * sr + s] += needle
    hay[40 * sr:40 * sr + s] += np.float32(0.2) * needle
    hay[45 * sr:75 * sr] += np.linspace(0.0, 0.2, 30 * sr).astype(np.float32)
    p = gpu.Config(chunk_size_s=30.0, overlap_length_s=1.0, distance_s=5.0, prominence=0.13).params(sr, gpu.Scale.LIB)
    p.min_distance = 480 * sr                                  # find_peaks' distance: longer than a chunk, as in matcher/args.rs:73-76
    algo = gpu.HipConvolve(needle)
    res = {}
    for order in (0, 1):
        exp = oracle.calc_chunks(sr, hay, needle, p.chunk, p.overlap, 0.13, p.min_distance, 5.0, pol=oracle.policy(order))
        with policy_set(gpu, peak_filter_order=order):
            for _ in range(2):
                got = algo.match(hay, p)
                assert [g.start for g in got] == [e[0] for e in exp], order
                for g, e in zip(got, exp):
                    assert abs(g.height - e[2]) < 1e-4 and abs(g.prominence - e[3]) < 1e-4
        res[order] = [e[0] for e in exp]
    assert 40 * sr in res[0] and 40 * sr not in res[1] and 10 * sr in res[1] and len(res[1]) == 2
