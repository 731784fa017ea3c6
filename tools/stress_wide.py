#!/usr/bin/env python3
"""Repeat a host-buffer correlation on the N = 2^22 plan with a fresh needle handle every
round and compare every result bit for bit with the first one."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
import audiomatch_amd as am   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(3001)
s = 900_001
needle = rng.uniform(-0.25, 0.25, s).astype(np.float32)
hay = rng.uniform(-0.25, 0.25, s + 3_000_000).astype(np.float32)
hay[1_000_000:1_000_000 + s] += needle
other = rng.uniform(-0.25, 0.25, 16_000_000).astype(np.float32)
ref = None
bad = 0
for it in range(iters):
    a = am.HipConvolve(needle)
    if it % 2:   # interleave a different, larger transfer so that buffers are really rewritten
        cfg = am.Config(chunk_size_s=100.0, overlap_length_s=s / 44100, distance_s=480.0, prominence=0.13)
        a.match(other, cfg.params(44100, am.Scale.LIB))
    got = a.correlate_with_sample(hay, am.Mode.Valid, True)
    a.close()
    if ref is None:
        ref = got
        direct = float(np.dot(hay[1_000_000:1_000_000 + s].astype(np.float64), needle.astype(np.float64)) / np.sum(needle.astype(np.float64) ** 2))
        print("plant score", got[1_000_000], "direct", direct, flush=True)
    elif not np.array_equal(got, ref):
        bad += 1
        d = np.nonzero(got != ref)[0]
        print("iter %d: %d differ, idx %d..%d, max %.3e" % (it, d.size, d[0], d[-1], np.abs(got[d] - ref[d]).max()), flush=True)
print("done: %d iterations, %d mismatches" % (iters, bad))
