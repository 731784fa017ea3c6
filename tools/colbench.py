#!/usr/bin/env python3
"""The 512-row column kernels on rows of 8192 points (N = 2^22, production) and of 16384 points (N = 2^23 = 512 x 16384):
time per launch and per transform point (am_debug_column_bench).  Equal numbers of points: 20 pairs of 2^22 against 10 of 2^23."""
import ctypes as C
import sys

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am

L = am.lib()
for rnd in range(2):
    for dense in (0, 1):
        for wide, pairs in ((0, 20), (1, 10), (0, 22), (1, 11)):
            a, b = C.c_double(0), C.c_double(0)
            am._check(L.am_debug_column_bench(0, wide, pairs, 30, dense, C.byref(a), C.byref(b)))
            pts = pairs * (1 << (23 if wide else 22))
            print("round %d %s rows of %5d, %2d pairs: K1 %.4f ms (%.3f ps/point)  K3 %.4f ms (%.3f ps/point)" %
                  (rnd, "dense " if dense else "sparse", 16384 if wide else 8192, pairs, a.value, a.value * 1e9 / pts, b.value, b.value * 1e9 / pts), flush=True)
