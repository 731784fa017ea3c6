#!/usr/bin/env python3
"""tests/test_gpu_random.py::test_random_wide_plan[0] and [1] alternated in one process: the
open intermittent item only ever shows in [1] when it follows [0] in the suite."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiomatch_amd as gpu   # noqa: E402
import pyoracle as oracle      # noqa: E402
import test_gpu_random as T     # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
extra = sys.argv[2] if len(sys.argv) > 2 else ""
gpu.gpu_identity = "stress"
bad = 0
for it in range(iters):
    for seed in (0, 1):
        if extra == "shutdown" and seed == 0 and it % 2 == 0:
            gpu.lib().am_shutdown()
        try:
            T.test_random_wide_plan(gpu, oracle, seed)
            print("iter", it, "seed", seed, "ok", flush=True)
        except AssertionError as e:
            bad += 1
            print("iter", it, "seed", seed, "FAILED", str(e)[:1500], flush=True)
print("done:", iters, "iterations,", bad, "failures")
