"""CPU tests of the checker's policy switches (oracle/oracle.h orc_policy): the rules that no source or test
available offline pins -- find_peaks' filter order and distance rule, chunked's tail window, filter_surrounding's
neighbours -- each exist in both variants; the designed inputs of tests/policy_cases.py tell them apart, and the
scipy order is cross-checked against scipy itself."""
import numpy as np
import pytest

import policy_cases as pc


def starts(pk):
    return [p[0] for p in pk]


def test_defaults_are_the_zero_policy(oracle):
    for name, (make, prom, dist) in pc.PEAK_CASES.items():
        x = make()
        assert oracle.find_peaks(x, prom, dist) == oracle.find_peaks(x, prom, dist, pol=oracle.policy()), name


def test_filter_order_differs_when_the_tallest_maximum_is_not_prominent(oracle):
    x = pc.tall_but_not_prominent()
    n = x.size
    a = oracle.find_peaks(x, 0.13, 10 ** 9, pol=oracle.policy(peak_filter_order=0))
    b = oracle.find_peaks(x, 0.13, 10 ** 9, pol=oracle.policy(peak_filter_order=1))
    assert starts(a) == [n // 4] and abs(a[0][3] - 0.5) < 0.05      # prominence first: the hit
    assert b == []                                                   # distance first: the tallest crest wins, then fails
    # with the tallest maximum prominent both orders agree (the reference's usual case: one clean hit per chunk)
    y = x.copy()
    y[n // 2] += np.float32(2.0)
    assert oracle.find_peaks(y, 0.13, 10 ** 9, pol=oracle.policy(0)) == oracle.find_peaks(y, 0.13, 10 ** 9, pol=oracle.policy(1))


@pytest.mark.parametrize("case", ["ramp_with_hits_d400", "tone_and_drift_d120"])
def test_distance_first_order_is_scipys(oracle, case):
    """scipy.signal.find_peaks evaluates `distance` before `prominence` (its documented order of conditions) with
    strict `<` between plateau middles: the checker's peak_filter_order = 1, distance_rule = 0 on plateau-free data."""
    from scipy.signal import find_peaks
    make, prom, dist = pc.PEAK_CASES[case]
    x = make()
    idx, props = find_peaks(x.astype(np.float64), distance=dist, prominence=prom)
    got = oracle.find_peaks(x, prom, dist, pol=oracle.policy(peak_filter_order=1))
    assert sorted(starts(got)) == [int(i) for i in idx]
    by_start = {p[0]: p for p in got}
    for i, pr in zip(idx, props["prominences"]):
        assert abs(by_start[int(i)][3] - pr) < 1e-6
    # and the default order keeps a different set on this input (that is what makes it a distinguishing vector)
    assert sorted(starts(oracle.find_peaks(x, prom, dist))) != [int(i) for i in idx]


def test_distance_rule_variants_on_plateaus(oracle):
    x = pc.plateaus()
    res = {r: sorted(starts(oracle.find_peaks(x, 0.05, 10, pol=oracle.policy(distance_rule=r)))) for r in range(4)}
    # middles, strict <: 12|22 are 10 apart (both kept), 34|44 are 10 apart (kept), 61|70: 9 (70 dropped), 70|81 ...
    assert res[0] == [10, 22, 31, 44, 60, 81, 100]
    # middles, <=: a gap of exactly 10 now drops the lower one
    assert res[1] == [10, 44, 60, 81, 100]
    # starts, strict <: 10|22 -> 12 apart (kept); 22|31 -> 9 (31 dropped); 60|70 -> 10 (kept)
    assert res[2] == [10, 22, 44, 60, 70, 81, 100]
    # starts, <=: 60|70 -> 10: dropped
    assert res[3] == [10, 22, 44, 60, 81, 100]
    assert len({tuple(v) for v in res.values()}) == 4


def test_tail_window_and_surrounding_variants(oracle):
    sr = 8000
    needle, hay, plants = pc.chunk_case(oracle.synth_uniform, sr)
    s = needle.size
    args = (sr, hay, needle, 30 * sr, s, 0.13, sr, 3.5)

    def run(**kw):
        return [p[0] // (sr // 10) / 10.0 for p in oracle.calc_chunks(*args, pol=oracle.policy(**kw))]

    assert run() == [5.0, 8.0, 40.0, 93.0]                       # defaults: 0.8 dropped (two stronger neighbours), 0.9 kept (before = 0.8)
    assert run(tail_window=1) == [5.0, 8.0, 40.0]                 # the hit in the short last window is not looked at
    assert run(surrounding_from=1) == [5.0, 40.0, 93.0]           # sequential filter: 0.9 meets the kept 1.0, 3 s away
    assert oracle.calc_chunks(*args) == oracle.calc_chunks(*args, pol=oracle.policy())
