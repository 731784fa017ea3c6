#!/usr/bin/env python3
"""Qualification of the half-precision levels (option half_pipeline; BASELINE configs[4]): EVERY score of a
10 s needle vs 1 h haystack correlation (am_correlate_device: the K1 / K2 / K3 the match path runs) against
an f64 overlap-add reference (scipy), for f32 and both half levels, on
  * the headline workload (white noise, six planted needles),
  * the three non-white signals of bench.py (tone + drift, AR(1), speech-like envelope),
  * a dynamic-range torture signal: a full-scale passage, a passage 60 dB down, a planted needle in each.
Prints one JSON object: max / rms / 99.99th-percentile absolute error per signal and level, and the error at
the planted hits.  The contract of the library is 1e-4 absolute (north_star); level 1 is meant to stay
inside it, level 2 is offsets-only."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
from scipy.signal import oaconvolve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
sys.path.insert(0, ROOT)
import audiomatch_amd as am   # noqa: E402
import bench                  # noqa: E402

SR = 44100
S = 10 * SR
H = (int(sys.argv[1]) if len(sys.argv) > 1 else 3600) * SR
DEV = 0


def white():
    needle = am.synth_uniform_device(DEV, S, seed=1, stream=0)
    hay = am.synth_uniform_device(DEV, H, seed=1, stream=1)
    plants = [t for t in bench.plant_offsets(0) if t + S <= H]
    for t in plants:
        am.axpy_device(DEV, hay, t, needle.ptr, S, 1.0)
    return needle.to_numpy(np.float32, S), hay.to_numpy(np.float32, H), plants


def from_bench(maker):
    def make():
        nbuf, algo, hbuf, plants, _ = maker(am, DEV, S, H) if H == 3600 * SR else maker(am, DEV, S, H)
        n, h = nbuf.to_numpy(np.float32, S), hbuf.to_numpy(np.float32, H)
        algo.close(); hbuf.free(); nbuf.free()
        return n, h, [t for t in plants if t + S <= H]
    return make


def dynamic_range():
    rng = np.random.default_rng(23)
    needle = rng.uniform(-0.25, 0.25, S).astype(np.float32)
    hay = rng.uniform(-1.0, 1.0, H).astype(np.float32)          # full scale ...
    hay[H // 2:] *= np.float32(0.001)                             # ... then 60 dB down
    plants = [H // 4, H // 2 + H // 4]
    hay[plants[0]:plants[0] + S] += needle
    hay[plants[1]:plants[1] + S] += needle * np.float32(0.001)    # a hit at the level of its passage
    return needle, hay, plants


SIGNALS = {"headline_white": white, "tone_and_drift": from_bench(bench.make_tonal), "ar1": from_bench(bench.make_ar1),
           "speechlike": from_bench(bench.make_speechlike), "full_scale_then_minus_60_dB": dynamic_range}


def main():
    out = {"needle_samples": S, "haystack_samples": H, "reference": "scipy.signal.oaconvolve in f64, divided by sum(needle^2)",
           "signals": {}}
    J = H - S + 1
    for name, make in SIGNALS.items():
        needle, hay, plants = make()
        t0 = time.time()
        e = float(np.sum(needle.astype(np.float64) ** 2))
        ref = oaconvolve(hay.astype(np.float64), needle.astype(np.float64)[::-1], mode="valid") / e
        print(f"{name}: host reference {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
        nb = am.DeviceBuffer.from_numpy(DEV, needle)
        hb = am.DeviceBuffer.from_numpy(DEV, hay)
        dst = am.DeviceBuffer(DEV, 4 * J)
        res = {"score_range": [float(ref.min()), float(ref.max())], "score_rms": float(np.sqrt(np.mean(ref ** 2)))}
        for level in (0, 1, 2):
            algo = am.HipConvolve.from_device(DEV, nb.ptr, S)
            algo.set_option("half_pipeline", level)
            n = C.c_size_t(0)
            am._check(am.lib().am_correlate_device(algo._h, hb.ptr, H, int(am.Mode.Valid), int(am.Scale.LIB), dst.ptr, J, C.byref(n)))
            got = dst.to_numpy(np.float32, J).astype(np.float64)
            err = np.abs(got - ref)
            res[f"level{level}"] = {"max_abs_err": float(err.max()), "rms_err": float(np.sqrt(np.mean(err ** 2))),
                                    "p9999_abs_err": float(np.quantile(err[::7], 0.9999)),
                                    "count_above_1e-4": int(np.sum(err > 1e-4)),
                                    "err_at_hits": [float(err[t]) for t in plants], "ref_at_hits": [float(ref[t]) for t in plants],
                                    "argmax_is_a_hit": bool(int(np.argmax(got)) in plants)}
            algo.close()
            del got, err
        out["signals"][name] = res
        nb.free(); hb.free(); dst.free()
        del ref, hay, needle
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
