import sys, time
sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am
SR=44100; h=3600*SR
hay = am.synth_uniform_device(0, h, 1, 1)
for secs in (0.5, 2, 5, 10, 20, 30, 40, 60):
    s = int(secs*SR)
    needle = am.synth_uniform_device(0, s, 1, 0)
    algo = am.HipConvolve.from_device(0, needle.ptr, s)
    h2 = am.synth_uniform_device(0, h, 1, 2)
    for m in range(6): am.axpy_device(0, h2, 600*SR*m + 30*SR + 7, needle.ptr, s, 1.0)
    cfg = am.Config(chunk_size_s=60, overlap_length_s=secs, distance_s=480.0, prominence=0.13 if secs >= 2 else 0.5)
    p = cfg.params(SR, am.Scale.LIB)
    for i in range(3): r = algo.match_device(h2.ptr, h, p)
    ok = [q.start for q in r] == [600*SR*m + 30*SR + 7 for m in range(6)]
    t0=time.perf_counter()
    for i in range(10): algo.match_device(h2.ptr, h, p)
    dt=(time.perf_counter()-t0)/10
    print(f"needle {secs:5.1f} s: {dt*1e3:7.3f} ms per hour  {h/dt:.3e} samples/s  hits_ok={ok}")
    h2.free(); algo.close(); needle.free()
