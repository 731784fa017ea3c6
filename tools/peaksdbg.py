import sys
sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am
SR=44100; s=10*SR; h=3600*SR
needle = am.synth_uniform_device(0, s, 1, 0)
algo = am.HipConvolve.from_device(0, needle.ptr, s)
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(SR, am.Scale.LIB)
for name, plants in (("no hits", []), ("6 hits", [600*SR*m + 30*SR + 1234 for m in range(6)]), ("59 hits", [60*SR*m + 30*SR + 1234 for m in range(59)])):
    hay = am.synth_uniform_device(0, h, 1, 1)
    for t in plants: am.axpy_device(0, hay, t, needle.ptr, s, 1.0)
    for i in range(3): r = algo.match_device(hay.ptr, h, p)
    with am.Profile(0) as prof:
        for i in range(10): r = algo.match_device(hay.ptr, h, p)
        print(name, len(r), {k: round(prof.query(k)[0]/10, 4) for k in ("k3_cols_inv","tile_stats","peaks")})
    hay.free()
