#!/usr/bin/env python3
"""What does a launch boundary cost?  The same hours of audio as 1 h, 2 h and 2.9 h haystacks (22, 44 and 64 block
pairs per K1 / K2 / K3 launch), batches through am_match_batch_device, ms per HOUR of audio.
Usage: python3 tools/launch_size_probe.py [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audio-matcher_amd", "python"))
import audiomatch_amd as am

SR, NEEDLE_S = 44100, 10


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    s = NEEDLE_S * SR
    needle = am.synth_uniform_device(0, s, seed=1, stream=0)
    algo = am.HipConvolve.from_device(0, needle.ptr, s)
    params = am.Config(chunk_size_s=60, overlap_length_s=NEEDLE_S, distance_s=480.0, prominence=0.13).params(SR, am.Scale.LIB)
    out = {}
    for hours, count in ((1.0, 12), (2.0, 6), (2.9, 4), (1.0, 12), (2.0, 6)):
        h = int(hours * 3600) * SR
        bufs = []
        for k in range(count):
            b = am.DeviceBuffer(0, 4 * h)
            am._check(am.lib().am_synth_uniform_device(0, b.ptr, 1, k + 1, 0, h, 0.25))
            am.axpy_device(0, b, 1_000_000 + 1000 * k, needle.ptr, s, 1.0)
            bufs.append(b)
        ptrs, lens = [b.ptr for b in bufs], [h] * count
        algo.match_batch_device(ptrs, lens, params, cap_per_hay=16)
        best = None
        for _ in range(rounds):
            am._check(am.lib().am_device_synchronize(0))
            t0 = time.perf_counter()
            res = algo.match_batch_device(ptrs, lens, params, cap_per_hay=16)
            am._check(am.lib().am_device_synchronize(0))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert all([p.start for p in r] == [1_000_000 + 1000 * k] for k, r in enumerate(res))
        out.setdefault(f"{hours}h_x{count}", []).append(round(best / (hours * count) * 1e3, 4))
        for b in bufs:
            b.free()
    print(json.dumps({"ms_per_hour_of_audio": out}))


if __name__ == "__main__":
    main()
