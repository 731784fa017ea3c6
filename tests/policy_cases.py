"""Inputs on which the unpinned rules of the path give DIFFERENT answers (oracle/oracle.h orc_policy, the library's
options of the same names): whoever can run the crates find_peaks 0.1 / common on them pins each rule with one run.
Used by tests/test_policy_variants.py (checker alone, CPU) and tests/test_gpu_policy.py (library == checker under every
combination)."""
import itertools

import numpy as np

PEAK_POLICIES = list(itertools.product((0, 1), (0, 1, 2, 3)))   # (peak_filter_order, distance_rule)


def tall_but_not_prominent(n=4000, seed=5):
    """A score array whose tallest maximum FAILS the prominence bound: a rising ramp with a small ripple (every ripple
    crest is a maximum of prominence ~0.02, the last one the tallest) and, lower down, one clean hit of prominence 0.5.  min_prominence 0.13,
    min_distance >= n (the reference's default regime):
      prominence -> distance (default): the hit survives;   distance -> prominence (scipy): the tallest crest
      suppresses everything and is then dropped itself: NO peak."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 1.0, n).astype(np.float32)
    x += (0.01 * np.sin(np.arange(n) * 0.9)).astype(np.float32) + (0.001 * rng.standard_normal(n)).astype(np.float32)
    x[n // 4 - 40:n // 4 + 41] -= np.float32(0.3)       # a dip either side of the hit makes it prominent
    x[n // 4] += np.float32(0.5)
    # (the ramp rises to the end of the array: its last ripple crest is the tallest maximum, prominence ~0.02)
    return x


def tone_and_drift(n=60000, seed=9):
    """bench.py's non-white signal in small: a slow drift (amplitude 0.13) under a ripple (0.03) and noise; many
    maxima pass the height test, few the prominence test.  With a min_distance shorter than the array the two filter
    orders keep different sets."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    x = 0.13 * np.sin(2 * np.pi * t / 24000.0) + 0.03 * np.sin(2 * np.pi * t / 100.0) + 0.004 * rng.standard_normal(n)
    for p, g in ((7000, 0.5), (7150, 0.6), (30000, 0.45), (30090, 0.40), (52000, 0.7)):
        x[p] += g
    return x.astype(np.float32)


def ramp_with_hits(n=8000, seed=9):
    """A steep ramp (0.4 per 400 scores) under a ripple, with five clean hits of prominence ~0.33, 1200+ apart: every
    ripple crest 300..400 scores up the ramp from a hit is TALLER than the hit and not prominent.  min_distance 400:
    prominence -> distance keeps all five hits; distance -> prominence lets a crest suppress the hit at 900 before the
    crest itself is dropped (scipy.signal.find_peaks agrees with the latter: test_distance_first_order_is_scipys)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    x = 1e-3 * t + 0.03 * np.sin(2 * np.pi * t / 100.0) + 0.002 * rng.standard_normal(n)
    for p in (900, 2100, 3350, 4800, 6500):
        x[p] += 0.3
    return x.astype(np.float32)


def plateaus():
    """Flat tops of even and odd length at spacings around min_distance = 10: the `<` / `<=` rule and the start / middle
    rule each change which ones survive (middles 10.. vs starts; a gap of exactly 10)."""
    x = np.zeros(120, dtype=np.float32)
    def top(a, b, h):
        x[a:b] = h
    top(10, 14, 1.00)     # start 10, end 14, middle 12
    top(22, 23, 0.90)     # start 22, middle 22: 10 from the middle of the first, 12 from its start
    top(31, 37, 0.80)     # middle 34, start 31
    top(44, 45, 0.95)     # middle 44: exactly 10 from 34
    top(60, 62, 0.70)     # middle 61
    top(70, 71, 0.60)     # 9 from 61, 10 from 60
    top(81, 82, 0.65)     # 11 / 10
    top(100, 101, 0.50)
    return x


PEAK_CASES = {
    # name: (array builder, min_prominence, min_distance)
    "tall_but_not_prominent": (tall_but_not_prominent, 0.13, 10 ** 9),
    "ramp_with_hits_d400": (ramp_with_hits, 0.13, 400),
    "tone_and_drift_d120": (tone_and_drift, 0.05, 120),
    "plateaus_d10": (plateaus, 0.05, 10),
}


def chunk_case(synth_uniform, sr=8000):
    """calc_chunks inputs (chunk 30 s, overlap = needle = 1 s, min_distance 1 s, overshadow distance 3.5 s) for the
    tail-window and the neighbour rule:
      * a hit at 93 s lies in the LAST window (90 s ..), which is shorter than chunk + overlap: tail_window = emit
        finds it, drop does not;
      * hits of strength 1.0, 0.8, 0.9 at 5.0, 6.5, 8.0 s: the 0.8 is overshadowed by the 1.0 either way; the 0.9 has
        the weaker 0.8 as its neighbour in the sorted UNFILTERED sequence (kept), but the 1.0 -- 3.0 s away -- as the
        last element KEPT (dropped by the sequential filter)."""
    s = sr
    needle = synth_uniform(41, 0, 0, s)
    hay = synth_uniform(41, 1, 0, 95 * sr + 1234)
    plants = [(5.0, 1.0), (6.5, 0.8), (8.0, 0.9),        # the chain (overshadow distance 3.5 s: 1.0 reaches 0.9 through the dropped 0.8)
              (40.0, 1.0),
              (93.0, 1.0)]                                 # in the short last window (windows start at 0, 30, 60, 90 s)
    for t, g in plants:
        off = int(t * sr)
        hay[off:off + s] += np.float32(g) * needle
    return needle, hay, plants
