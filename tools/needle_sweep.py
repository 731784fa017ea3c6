#!/usr/bin/env python3
"""ms per hour of audio for needles of 0.5 .. 120 s, on the automatic plan and on forced
N = 2^21 / 2^22 / 2^23 (plan policy of am_api.hip pick_log_n).  Warmed-up clocks."""
import sys
import time

sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am

SR = 44100; h = 3600 * SR
secs_list = [float(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0.5, 2, 5, 8, 10, 12, 16, 20, 30, 40, 60, 90, 120]
for secs in secs_list:
    s = int(secs * SR)
    needle = am.synth_uniform_device(0, s, 1, 0)
    algo = am.HipConvolve.from_device(0, needle.ptr, s)
    h2 = am.synth_uniform_device(0, h, 1, 2)
    plants = [600 * SR * m + 30 * SR + 7 for m in range(6)]
    for t in plants:
        am.axpy_device(0, h2, t, needle.ptr, s, 1.0)
    cfg = am.Config(chunk_size_s=60, overlap_length_s=secs, distance_s=480.0, prominence=0.13 if secs >= 2 else 0.5)
    p = cfg.params(SR, am.Scale.LIB)
    row = []
    for log_n in (0, 21, 22, 23):
        if log_n and (1 << log_n) < 2 * s:
            row.append("   -   ")
            continue
        am.set_option("log_n", log_n)
        for _ in range(30):
            r = algo.match_device(h2.ptr, h, p)
        ok = [q.start for q in r] == plants
        t0 = time.perf_counter()
        for _ in range(30):
            algo.match_device(h2.ptr, h, p)
        dt = (time.perf_counter() - t0) / 30
        row.append("%6.3f%s" % (dt * 1e3, "" if ok else "!"))
    am.set_option("log_n", 0)
    print("needle %5.1f s (%8d samples): auto %s   2^21 %s   2^22 %s   2^23 %s  ms per hour" % (secs, s, row[0], row[1], row[2], row[3]), flush=True)
    h2.free(); algo.close(); needle.free()
