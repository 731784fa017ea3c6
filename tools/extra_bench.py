#!/usr/bin/env python3
"""Extra measurements quoted in DESIGN.md (not the driver's bench line): batch of haystacks
(BASELINE config 3 shape per GPU), multi-needle (config 4 shape), 48 kHz i16 stereo ingest with
the half-precision work matrix (config 5 shape)."""
import json
import sys
import time

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am
import numpy as np

dev = 0
out = {}
SR = 44100; s = 10 * SR; h = 3600 * SR
needle = am.synth_uniform_device(dev, s, 1, 0)
algo = am.HipConvolve.from_device(dev, needle.ptr, s)
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(SR, am.Scale.LIB)

def plants(k): return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]

# ---- config 3 shape: a batch of resident haystacks through am_match_batch_device ----
nh = 16
hays = []
for k in range(nh):
    b = am.synth_uniform_device(dev, h, 1, k + 1)
    for t in plants(k): am.axpy_device(dev, b, t, needle.ptr, s, 1.0)
    hays.append(b)
algo.match_batch_device([b.ptr for b in hays], [h] * nh, p)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    res = algo.match_batch_device([b.ptr for b in hays], [h] * nh, p)
dt = (time.perf_counter() - t0) / reps
assert all([q.start for q in r] == plants(k) for k, r in enumerate(res))
out["batch16_resident"] = {"samples_per_s": nh * h / dt, "ms_per_haystack": dt / nh * 1e3}

# ---- config 4 shape: 8 needles vs one haystack, shared forward pass ----
nn = 8
needles = [am.synth_uniform_device(dev, s, 1, 2001 + k) for k in range(nn)]
algos = [am.HipConvolve.from_device(dev, n.ptr, s) for n in needles]
hay = am.synth_uniform_device(dev, h, 1, 1)
for k, n in enumerate(needles):
    for t in (310 * SR + 1000 * k, 2010 * SR + 999 * k): am.axpy_device(dev, hay, t, n.ptr, s, 1.0)
for _ in range(5): am.match_multi_device(algos, hay.ptr, h, p)   # clock ramp
t0 = time.perf_counter()
for _ in range(reps):
    res = am.match_multi_device(algos, hay.ptr, h, p)
dt = (time.perf_counter() - t0) / reps
assert all([q.start for q in r] == [310 * SR + 1000 * k, 2010 * SR + 999 * k] for k, r in enumerate(res))
t0 = time.perf_counter()
for _ in range(reps):
    for a in algos: a.match_device(hay.ptr, h, p)
dt1 = (time.perf_counter() - t0) / reps
out["multi_needle8"] = {"needle_samples_per_s": nn * h / dt, "separate_calls_needle_samples_per_s": nn * h / dt1}

# ---- config 5 shape: 48 kHz interleaved i16 stereo, f32 and half-precision work matrix ----
# a resident batch (as the headline bench runs the f32 mono case), clock ramp, per-kernel breakdown
SR5 = 48000; s5 = 10 * SR5; h5 = 3600 * SR5; nb5 = 4
KN = ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks")
rng = np.random.default_rng(1)
nl = rng.integers(-8000, 8000, size=2 * s5).astype(np.int16)
pl5 = [600 * SR5 * m + 30 * SR5 for m in range(6)]
bufs5 = []
for k in range(nb5):
    hl = rng.integers(-8000, 8000, size=2 * h5).astype(np.int16)
    for t in pl5:
        seg = hl[2 * t:2 * (t + s5)].astype(np.int32) + nl
        hl[2 * t:2 * (t + s5)] = np.clip(seg, -32768, 32767).astype(np.int16)
    bufs5.append(am.DeviceBuffer.from_numpy(dev, hl))
    del hl
cfg5 = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p5 = cfg5.params(SR5, am.Scale.LIB)
ptrs5 = [b.ptr for b in bufs5]
for mode in (0, 1, 2):
    am.set_option("half_pipeline", mode)
    a5 = am.HipConvolve.from_pcm16(nl)
    for _ in range(12): r = a5.match_pcm16_batch_device(ptrs5, [h5] * nb5, p5)
    assert all([q.start for q in one] == pl5 for one in r), r
    t0 = time.perf_counter()
    for _ in range(10): r = a5.match_pcm16_batch_device(ptrs5, [h5] * nb5, p5)
    dt = (time.perf_counter() - t0) / (10 * nb5)
    am.set_option("profile_mask", -1)
    with am.Profile(dev) as prof:
        for _ in range(3): a5.match_pcm16_batch_device(ptrs5, [h5] * nb5, p5)
        kern = {n: round(prof.query(n)[0] / (3 * nb5), 4) for n in KN}
    out["pcm16_48k_" + ("f32", "half", "half_f16_butterflies")[mode]] = {"samples_per_s": h5 / dt, "ms_per_hour": dt * 1e3,
                                                       "kernel_ms_per_hour": kern}
am.set_option("half_pipeline", 0)
print(json.dumps(out, indent=1))
