import sys, time
sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am
SR = 44100; h = 3600 * SR
for secs in (60, 120):
    s = int(secs * SR)
    needle = am.synth_uniform_device(0, s, 1, 0)
    algo = am.HipConvolve.from_device(0, needle.ptr, s)
    h2 = am.synth_uniform_device(0, h, 1, 2)
    p = am.Config(chunk_size_s=60, overlap_length_s=secs, distance_s=480.0, prominence=0.13).params(SR, am.Scale.LIB)
    for _ in range(5): algo.match_device(h2.ptr, h, p)
    am.set_option("profile_mask", -1)
    with am.Profile(0) as prof:
        t0 = time.perf_counter()
        for _ in range(5): algo.match_device(h2.ptr, h, p)
        dt = (time.perf_counter() - t0) / 5
        print(secs, "ms/call", dt * 1e3, {k: (round(prof.query(k)[0] / 5, 3), prof.query(k)[1] // 5) for k in ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks", "other")}, flush=True)
    h2.free()
