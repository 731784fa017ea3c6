#!/usr/bin/env python3
"""BASELINE config 3 shape on one GPU: a batch of resident 1 h haystacks through
am_match_batch_device, with and without the overlapped peak pick."""
import json
import sys
import time

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am

dev = 0
SR = 44100; s = 10 * SR; h = 3600 * SR
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 16
needle = am.synth_uniform_device(dev, s, 1, 0)
algo = am.HipConvolve.from_device(dev, needle.ptr, s)
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(SR, am.Scale.LIB)


def plants(k):
    return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]


hays = []
for k in range(nh):
    b = am.synth_uniform_device(dev, h, 1, k + 1)
    for t in plants(k):
        am.axpy_device(dev, b, t, needle.ptr, s, 1.0)
    hays.append(b)
ptrs, lens = [b.ptr for b in hays], [h] * nh
out = {}
for mode in (0, 1, 0, 1):
    am.set_option("batch_overlap", mode)
    for _ in range(8):
        res = algo.match_batch_device(ptrs, lens, p)      # clock ramp
    assert all([q.start for q in r] == plants(k) for k, r in enumerate(res))
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        algo.match_batch_device(ptrs, lens, p)
    dt = (time.perf_counter() - t0) / reps
    out.setdefault("overlap%d" % mode, []).append({"samples_per_s": nh * h / dt, "ms_per_haystack": dt / nh * 1e3})
print(json.dumps(out, indent=1))
