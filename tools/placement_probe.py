#!/usr/bin/env python3
"""Is the K1/K3 time sensitive to where the buffers land?  Allocates a dummy buffer of the given
size first (shifting every later allocation), then runs the headline loop and prints the per-kernel
times and the addresses of the haystack buffers."""
import sys
import time

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am

pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SR = 44100; s = 10 * SR; h = 3600 * SR
dummy = am.DeviceBuffer(0, pad) if pad else None
needle = am.synth_uniform_device(0, s, 1, 0)
algo = am.HipConvolve.from_device(0, needle.ptr, s)
hay = am.synth_uniform_device(0, h, 1, 1)
for m in range(6):
    am.axpy_device(0, hay, 600 * SR * m + 30 * SR + 1234, needle.ptr, s, 1.0)
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(SR, am.Scale.LIB)
for _ in range(150):
    algo.match_device(hay.ptr, h, p)
t0 = time.perf_counter()
for _ in range(200):
    algo.match_device(hay.ptr, h, p)
dt = (time.perf_counter() - t0) / 200
KN = ("k1_cols_fwd", "k2_rows", "k3_cols_inv")
with am.Profile(0) as prof:
    for _ in range(20):
        algo.match_device(hay.ptr, h, p)
    kern = {n: round(prof.query(n)[0] / 20, 4) for n in KN}
print("pad %10d  hay %#x needle %#x  ms/step %.4f  %s" % (pad, hay.ptr, needle.ptr, dt * 1e3, kern))
