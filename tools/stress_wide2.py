#!/usr/bin/env python3
"""The body of tests/test_gpu_random.py::test_random_wide_plan[seed], repeated in one process,
with the positions of any score that is off by more than 1e-5."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiomatch_amd as gpu   # noqa: E402
import pyoracle as oracle      # noqa: E402
from test_gpu_random import build_case, compare   # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(3000 + seed)
sr = 44100
s = int(rng.integers(600_000, 1_200_000))
h = int(rng.integers(14_000_000, 18_000_000))
needle, hay = build_case(oracle, rng, sr, s, h, int(rng.integers(1, 4)))
chunk = int(rng.integers(4_000_000, 6_000_000)) | 1
off = chunk + int(rng.integers(1, 40))
hay[off:off + s] += needle
dist = float(rng.choice([5.0, 480.0]))
win = hay[: s + 3_000_000]
ref = oracle.correlate(win, needle, oracle.MODE_VALID, oracle.SCALE_LIB)
print("s", s, "h", h, "chunk", chunk, flush=True)
first = None
for it in range(iters):
    compare(gpu, oracle, needle, hay, sr, chunk, s, 0.13, dist)
    got = gpu.HipConvolve(needle).correlate_with_sample(win, gpu.Mode.Valid, True)
    err = np.abs(got - ref)
    bad = np.nonzero(err > 1e-5)[0]
    same = first is None or np.array_equal(got, first)
    if first is None:
        first = got
    print("iter", it, "max err %.3e" % err.max(), "n>1e-5:", bad.size, "same as first:", same,
          ("idx %d..%d" % (bad[0], bad[-1])) if bad.size else "", flush=True)
    if bad.size:
        print("   sample idx:", bad[:16], " mod 8192:", np.unique(bad % 8192)[:20], " //16384:", np.unique(bad // 16384)[:20], flush=True)
