import sys
sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd")
import audiomatch_amd as am
SR=44100; s=10*SR; h=3600*SR
needle = am.synth_uniform_device(0, s, 1, 0)
algo = am.HipConvolve.from_device(0, needle.ptr, s)
hay = am.synth_uniform_device(0, h, 1, 1)
cfg = am.Config(chunk_size_s=60, overlap_length_s=10, distance_s=480.0, prominence=0.13)
p = cfg.params(SR, am.Scale.LIB)
for i in range(3): algo.match_device(hay.ptr, h, p)
for dbg, name in ((0,"full"), (1,"no Hc load"), (2,"no twiddles"), (3,"no barriers"), (0,"full")):
    am.set_option("k2_debug", dbg)
    with am.Profile(0) as prof:
        for i in range(10):
            try: algo.match_device(hay.ptr, h, p)
            except Exception as e: pass
        print(name, {k: round(prof.query(k)[0]/10, 3) for k in ("k1_cols_fwd","k2_rows","k3_cols_inv")})
