#!/usr/bin/env python3
"""Alternate two different inputs through the same handle and buffers on small (one-pair)
problems, whose whole working set can sit in the L2s, and compare every result bit for bit with
the first result for that input: a stale cache line or a lost store between kernels would show
as pieces of the other input's data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
import audiomatch_amd as am   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(77)
cases = {"wide 2^22 (1 pair)": (900_001, 3_000_000), "r16 2^21 (1 pair)": (300_001, 1_500_000),
         "r16 2^21 (2 pairs)": (300_001, 5_000_000), "generic 2^17": (20_001, 90_000)}
for name, (s, extra) in cases.items():
    needle = rng.uniform(-0.25, 0.25, s).astype(np.float32)
    ins = [rng.uniform(-0.25, 0.25, s + extra).astype(np.float32) for _ in range(2)]
    a = am.HipConvolve(needle)
    ref = [None, None]
    bad = 0
    for it in range(iters):
        k = it & 1
        got = a.correlate_with_sample(ins[k], am.Mode.Valid, True)
        if ref[k] is None:
            ref[k] = got
        elif not np.array_equal(got, ref[k]):
            bad += 1
            d = np.nonzero(got != ref[k])[0]
            print("  %s iter %d input %d: %d differ, idx %d..%d, max %.3e, cols(mod 16384) %s" %
                  (name, it, k, d.size, d[0], d[-1], np.abs(got[d] - ref[k][d]).max(), np.unique(d % 16384)[:12]), flush=True)
    a.close()
    print("%-22s %d iterations, %d mismatches" % (name, iters, bad), flush=True)
