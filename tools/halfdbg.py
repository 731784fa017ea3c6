import sys
sys.path.insert(0, "audio-matcher_amd/python"); sys.path.insert(0, "audio-matcher_amd"); sys.path.insert(0, "oracle")
import numpy as np, audiomatch_amd as am, pyoracle as po
sr=44100; s=10*sr
needle = po.synth_uniform(1,0,0,s); hay = po.synth_uniform(1,1,0,3*1024*1024)
hay[1000000:1000000+s] += needle
ref = po.correlate(hay, needle, po.MODE_VALID, po.SCALE_LIB)
algo = am.HipConvolve(needle)
am.set_option("log_n", 21)
full = algo.correlate_with_sample(hay, am.Mode.Valid, True)
am.set_option("half_pipeline", 1)
half = algo.correlate_with_sample(hay, am.Mode.Valid, True)
print("full err", np.abs(full-ref).max(), "half err", np.abs(half-ref).max())
print("argmax", ref.argmax(), full.argmax(), half.argmax())
i=1000000
print(ref[i-2:i+4]); print(half[i-2:i+4])
# swapped pairs?
sw = half.copy(); n=(len(sw)//2)*2; sw[:n:2], sw[1:n:2] = half[1:n:2].copy(), half[:n:2].copy()
print("pair-swapped err", np.abs(sw[:n]-ref[:n]).max())
