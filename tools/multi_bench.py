#!/usr/bin/env python3
"""BASELINE config 4 shape on one GPU: NN needles against one resident 1 h haystack through
am_match_multi_device, for several needle_group settings, with the per-kernel breakdown."""
import json
import sys
import time

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am

dev = 0
SR = 44100; s = 10 * SR; h = 3600 * SR
lean = "--lean" in sys.argv          # a few calls only (under rocprofv3 --pmc every kernel runs serialised)
argv = [a for a in sys.argv if a != "--lean"]
nn = int(argv[1]) if len(argv) > 1 else 8
groups = [int(x) for x in argv[2].split(",")] if len(argv) > 2 else [1, 2, 4, 8]
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(SR, am.Scale.LIB)
needles = [am.synth_uniform_device(dev, s, 1, 2001 + k) for k in range(nn)]
algos = [am.HipConvolve.from_device(dev, n.ptr, s) for n in needles]
hay = am.synth_uniform_device(dev, h, 1, 1)
for k, n in enumerate(needles):
    for t in (310 * SR + 1000 * k, 2010 * SR + 999 * k):
        am.axpy_device(dev, hay, t, n.ptr, s, 1.0)
out = {}
KN = ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks")
for g in groups:
    am.set_option("needle_group", g)
    for _ in range(2 if lean else 12):
        res = am.match_multi_device(algos, hay.ptr, h, p)     # clock ramp + sparse-score state
    assert all([q.start for q in r] == [310 * SR + 1000 * k, 2010 * SR + 999 * k] for k, r in enumerate(res))
    reps = 2 if lean else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        am.match_multi_device(algos, hay.ptr, h, p)
    dt = (time.perf_counter() - t0) / reps
    with am.Profile(dev) as prof:
        for _ in range(3):
            am.match_multi_device(algos, hay.ptr, h, p)
        kern = {n: round(prof.query(n)[0] / 3, 4) for n in KN}
    rate = nn * h / dt
    # the bytes this design moves per call (bench.py's MultiNeedleWorkload): K1 once (two f32 blocks in, one complex
    # point out), per group of g needles the rows read once, written g times and g spectra, K3 once per needle
    npairs, n_fft = 22, 1 << 22
    pts = npairs * n_fft
    ngroups = -(-nn // g)
    design = pts * 16 + (ngroups * pts * 8 + nn * (pts * 8 + n_fft * 8)) + nn * (pts * 8 + (h - s + 1) // 4)
    # (16 + 16 K) N bytes per block of N - S + 1 samples at K = 32, N = 2^22: 18.44 B per needle-sample
    out[f"group{g}"] = {"needles": nn, "needle_samples_per_s": rate, "ms_per_needle_hour": dt / nn * 1e3,
                        "design_bytes_per_call": design, "design_frac_of_8TBs": design / dt / 8e12,
                        "survey_model_frac_of_8TBs": rate * 18.44 / 8e12, "kernel_ms_per_call": kern}
print(json.dumps(out, indent=1))
