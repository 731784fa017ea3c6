#!/usr/bin/env python3
"""Full-size score check: every score of the 10 s needle vs 1 h haystack correlation
(am_correlate_device, the same K1/K2/K3 the match path runs) against an f64
overlap-add correlation computed with scipy on the host.  Prints the largest
absolute error, where it is, and whether a second run is bit-identical."""
import ctypes as C
import os
import sys
import time

import numpy as np
from scipy.signal import oaconvolve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
import audiomatch_amd as am   # noqa: E402

SR = 44100
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10 * SR
H = int(sys.argv[2]) if len(sys.argv) > 2 else 3600 * SR


def plant_offsets(k):
    return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]


def main():
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        am.set_option(k, int(v))
    needle = am.synth_uniform_device(0, S, seed=1, stream=0)
    algo = am.HipConvolve.from_device(0, needle.ptr, S)
    hay = am.synth_uniform_device(0, H, seed=1, stream=1)
    for t in plant_offsets(0):
        if t + S <= H:
            am.axpy_device(0, hay, t, needle.ptr, S, 1.0)
    J = H - S + 1
    out = am.DeviceBuffer(0, 4 * J)
    n = C.c_size_t(0)

    def run():
        am._check(am.lib().am_correlate_device(algo._h, hay.ptr, H, int(am.Mode.Valid), int(am.Scale.LIB),
                                               out.ptr, J, C.byref(n)))
        return out.to_numpy(np.float32, J)

    a = run()
    b = run()
    print("bit-identical second run:", bool(np.array_equal(a, b)), flush=True)
    h_host = hay.to_numpy(np.float32, H).astype(np.float64)
    n_host = needle.to_numpy(np.float32, S).astype(np.float64)
    e = float(np.sum(n_host ** 2))
    print("inv_autocorr gpu %.9g  host %.9g" % (algo.inverse_sample_auto_correlation(), 1.0 / e), flush=True)
    t0 = time.time()
    ref = oaconvolve(h_host, n_host[::-1], mode="valid") / e
    print("host reference %.1f s" % (time.time() - t0), flush=True)
    err = np.abs(a.astype(np.float64) - ref)
    worst = int(np.argmax(err))
    print("max abs err %.3e at %d (gpu %.7f ref %.7f)" % (err[worst], worst, a[worst], ref[worst]))
    print("count err > 1e-4:", int(np.sum(err > 1e-4)), " > 1e-5:", int(np.sum(err > 1e-5)), " > 1e-6:", int(np.sum(err > 1e-6)))
    bad = np.nonzero(err > 1e-5)[0]
    if bad.size:
        print("first bad:", bad[:10], "last bad:", bad[-10:])
    for t in plant_offsets(0):
        if t < J:
            print("plant %d gpu %.7f ref %.7f" % (t, a[t], ref[t]))
    print("rms err %.3e" % float(np.sqrt(np.mean(err ** 2))))


if __name__ == "__main__":
    main()
