#!/usr/bin/env python3
"""The non-white signal of bench.py's side measurement on its own (for rocprofv3 --kernel-trace
--stats): drift + ripple score array, thousands of candidate maxima per chunk."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
import audiomatch_amd as am   # noqa: E402
import bench                  # noqa: E402

s, h = bench.NEEDLE_S * bench.SR, bench.HAY_S * bench.SR
nbuf, algo, hbuf, plants = bench.make_tonal(am, 0, s, h)
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(bench.SR, am.Scale.LIB)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    am.set_option(k, int(v))
res = algo.match_device(hbuf.ptr, h, p)
print("offsets ok:", [q.start for q in res] == plants, len(res))
t0 = time.perf_counter()
n = 5
for _ in range(n):
    algo.match_device(hbuf.ptr, h, p)
print("ms per haystack: %.3f" % ((time.perf_counter() - t0) / n * 1e3))
