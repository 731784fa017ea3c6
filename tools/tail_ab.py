#!/usr/bin/env python3
"""The odd last block on the smaller plan (option tail_block) against the plain layout, for haystack lengths whose
block count is 41, 43 (1 h), 45: 8 resident haystacks per call, wall clock per haystack and the HIP-event averages
of the main pass's three kernels.  Usage: tools/tail_ab.py [blocks ...]"""
import json
import sys
import time

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am

dev = 0
SR = 44100; s = 10 * SR
hop = ((1 << 22) - s + 1) // 1024 * 1024
nh = 8
needle = am.synth_uniform_device(dev, s, 1, 0)
algo = am.HipConvolve.from_device(dev, needle.ptr, s)
p = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13).params(SR, am.Scale.LIB)
am.set_option("profile_mask", -1)
out = []
for blocks in [int(a) for a in sys.argv[1:]] or [41, 43, 45]:
    h = (blocks - 1) * hop + 700000 + s - 1          # the last block holds 700 000 scores
    hays = []
    for k in range(nh):
        b = am.synth_uniform_device(dev, h, 1, k + 1)
        for t in (30 * SR + 17 * k, h - s - 3 * SR - k):
            am.axpy_device(dev, b, t, needle.ptr, s, 1.0)
        hays.append(b)
    ptrs, lens = [b.ptr for b in hays], [h] * nh
    row = {"blocks": blocks, "pairs": (blocks + 1) // 2, "haystack_s": h / SR}
    for mode in (0, 1, 0, 1):
        am.set_option("tail_block", mode)
        for _ in range(6):
            res = algo.match_batch_device(ptrs, lens, p)      # clock ramp
        assert all([q.start for q in r] == [30 * SR + 17 * k, h - s - 3 * SR - k] for k, r in enumerate(res)), mode
        reps = 12
        t0 = time.perf_counter()
        for _ in range(reps):
            algo.match_batch_device(ptrs, lens, p)
        dt = (time.perf_counter() - t0) / reps
        with am.Profile(dev) as prof:
            for _ in range(3):
                algo.match_batch_device(ptrs, lens, p)
            ks = {k_: round(prof.query(k_)[0] / max(prof.query(k_)[1], 1) * 1e3, 1) for k_ in ("k1_cols_fwd", "k2_rows", "k3_cols_inv")}
        row.setdefault("tail_block=%d" % mode, []).append({"ms_per_haystack": round(dt / nh * 1e3, 4), "main_pass_us": ks})
    out.append(row)
    del hays
am.set_option("tail_block", 1)
print(json.dumps(out, indent=1))
