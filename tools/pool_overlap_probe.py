#!/usr/bin/env python3
"""Would two pipelines side by side on ONE GPU beat one?  The pool (am_pool_*) with 1, 2 and 3 slots on
device 0: every slot has a context of its own (streams, work matrix, score sets) and a submit thread, so
the kernels of slot 0's haystack k run beside those of slot 1's haystack k + 1 -- K1 (memory-bound) beside
K3 (issue- and latency-bound), and every launch's tail filled by the other slot's kernels.
Usage: python3 tools/pool_overlap_probe.py [haystacks] [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audio-matcher_amd", "python"))
import numpy as np

import audiomatch_amd as am

SR, NEEDLE_S, HAY_S = 44100, 10, 3600


def main():
    n_hay = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    s, h = NEEDLE_S * SR, HAY_S * SR
    needle_dev = am.synth_uniform_device(0, s, seed=1, stream=0)
    needle = needle_dev.to_numpy(np.float32, s)
    cfg = am.Config(chunk_size_s=60, overlap_length_s=NEEDLE_S, distance_s=480.0, prominence=0.13)
    params = cfg.params(SR, am.Scale.LIB)
    bufs = []
    for k in range(n_hay):
        b = am.DeviceBuffer(0, 4 * h)
        am._check(am.lib().am_synth_uniform_device(0, b.ptr, 1, k + 1, 0, h, 0.25))
        am.axpy_device(0, b, 1_000_000 + 1000 * k, needle_dev.ptr, s, 1.0)
        bufs.append(b)
    ptrs, lens = [b.ptr for b in bufs], [h] * n_hay
    out = {}
    for slots in (1, 2, 3, 1, 2):
        pool = am.Pool(needle, [0] * slots)
        pool.match_batch_device(ptrs, lens, params, cap_per_hay=16)   # warm-up: plans, spectra, clocks
        best = None
        for _ in range(rounds):
            am._check(am.lib().am_device_synchronize(0))
            t0 = time.perf_counter()
            res = pool.match_batch_device(ptrs, lens, params, cap_per_hay=16)
            am._check(am.lib().am_device_synchronize(0))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert all([p.start for p in r] == [1_000_000 + 1000 * k] for k, r in enumerate(res)), "offsets"
        out.setdefault(f"slots_{slots}", []).append(round(best / n_hay * 1e3, 4))
        pool.close()
    print(json.dumps({"ms_per_haystack_best_of_%d" % rounds: out, "haystacks": n_hay}))


if __name__ == "__main__":
    main()
