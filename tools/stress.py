#!/usr/bin/env python3
"""Bit-for-bit repeatability loops (hunting for intermittent data races).

  stress.py [iters] [mix] [opt=val ...]   full-size correlation / match repeated, every result
                                          compared with the first one
  stress.py alt [iters]                   two different inputs alternated through one handle on
                                          one-pair problems of every plan family (whole working
                                          set in the L2s): a stale line or a lost store between
                                          kernels shows as pieces of the other input's data
  stress.py tail [iters]                  ragged batches of haystacks with and without an odd last
                                          block (option tail_block: tails several per launch, copied
                                          into the score sets on the pick's stream), dips that fail
                                          certificates in the tail's chunk, device redo armed; every
                                          batch result compared with the single calls'
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
import audiomatch_amd as am   # noqa: E402

SR = 44100
S = 10 * SR
H = 3600 * SR


def plant_offsets(k):
    return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]


def alternate(iters):
    rng = np.random.default_rng(77)
    cases = {"wide 2^22 (1 pair)": (900_001, 3_000_000), "r16 2^21 (1 pair)": (300_001, 1_500_000),
             "r16 2^21 (2 pairs)": (300_001, 5_000_000), "generic 2^17": (20_001, 90_000)}
    for name, (s, extra) in cases.items():
        needle = rng.uniform(-0.25, 0.25, s).astype(np.float32)
        ins = [rng.uniform(-0.25, 0.25, s + extra).astype(np.float32) for _ in range(2)]
        a = am.HipConvolve(needle)
        ref = [None, None]
        bad = 0
        for it in range(iters):
            k = it & 1
            got = a.correlate_with_sample(ins[k], am.Mode.Valid, True)
            if ref[k] is None:
                ref[k] = got
            elif not np.array_equal(got, ref[k]):
                bad += 1
                d = np.nonzero(got != ref[k])[0]
                print("  %s iter %d input %d: %d differ, idx %d..%d, max %.3e, cols(mod 16384) %s" %
                      (name, it, k, d.size, d[0], d[-1], np.abs(got[d] - ref[k][d]).max(), np.unique(d % 16384)[:12]), flush=True)
        a.close()
        print("%-22s %d iterations, %d mismatches" % (name, iters, bad), flush=True)


def tails(iters):
    rng = np.random.default_rng(5)
    s = S
    hop = ((1 << 22) - s + 1) // 1024 * 1024
    needle = am.synth_uniform_device(0, s, seed=3, stream=0)
    algo = am.HipConvolve.from_device(0, needle.ptr, s)
    p = am.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=2.0, prominence=0.13).params(SR, am.Scale.LIB)
    key = lambda r: [(q.start, q.end, q.height, q.prominence) for q in r]
    hays, lens = [], []
    for k, (blocks, rest) in enumerate([(2, 700000), (4, 1900000), (3, 2000000), (6, 3000000), (2, 3311616), (4, 100), (8, 1234567), (5, 50000)]):
        out = blocks * hop + rest if blocks % 2 == 0 else (blocks - 1) * hop + rest + hop      # odd `blocks`: an even block count
        n = out + s - 1
        b = am.synth_uniform_device(0, n, seed=3, stream=k + 1)
        for t in (7 * SR + k, out - 1 - 3 * SR - k, (blocks - blocks % 2) * hop + 5):
            am.axpy_device(0, b, t, needle.ptr, s, 1.0)
        if k % 3 == 0:                                                                       # a dip in the last chunk
            am.axpy_device(0, b, out - 1 - 20 * SR, needle.ptr, s, -1.0)
        hays.append(b); lens.append(n)
    want = [key(algo.match_device(b.ptr, n, p)) for b, n in zip(hays, lens)]
    print("single calls:", [len(w) for w in want], flush=True)
    bad = 0
    for it in range(iters):
        order = list(rng.permutation(len(hays))) + list(rng.integers(0, len(hays), size=int(rng.integers(0, 9))))
        am.set_option("debug_redo_arm_at", int(rng.integers(-2, 3)))
        res = algo.match_batch_device([hays[i].ptr for i in order], [lens[i] for i in order], p)
        for j, i in enumerate(order):
            if key(res[j]) != want[i]:
                bad += 1
                print("iter %d: haystack %d at place %d of %s differs: %s != %s" % (it, i, j, order, key(res[j]), want[i]), flush=True)
        if it % 20 == 0:
            print("iter", it, "bad so far", bad, flush=True)
    am.set_option("debug_redo_arm_at", -2)
    print("done: %d batches, %d mismatches" % (iters, bad))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "alt":
        return alternate(int(sys.argv[2]) if len(sys.argv) > 2 else 300)
    if len(sys.argv) > 1 and sys.argv[1] == "tail":
        return tails(int(sys.argv[2]) if len(sys.argv) > 2 else 100)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    mix = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        am.set_option(k, int(v))
    needle = am.synth_uniform_device(0, S, seed=1, stream=0)
    algo = am.HipConvolve.from_device(0, needle.ptr, S)
    hay = am.synth_uniform_device(0, H, seed=1, stream=1)
    for t in plant_offsets(0):
        am.axpy_device(0, hay, t, needle.ptr, S, 1.0)
    cfg = am.Config(chunk_size_s=60.0, overlap_length_s=10.0, distance_s=480.0, prominence=0.13)
    p = cfg.params(SR, am.Scale.LIB)
    J = H - S + 1
    out = am.DeviceBuffer(0, 4 * J)
    n = C.c_size_t(0)

    def corr():
        am._check(am.lib().am_correlate_device(algo._h, hay.ptr, H, int(am.Mode.Valid), int(am.Scale.LIB),
                                               out.ptr, J, C.byref(n)))
        return out.to_numpy(np.float32, J)

    key = lambda r: [(q.start, q.end, q.height, q.prominence) for q in r]
    ref = corr()
    mref = key(algo.match_device(hay.ptr, H, p))
    bad = 0
    for it in range(iters):
        if mix:
            m = key(algo.match_device(hay.ptr, H, p))
            if m != mref:
                bad += 1
                print("iter %d: match differs: %s" % (it, m), flush=True)
            if mix > 1:
                k = 12345
                algo.match_device(hay.ptr + 4 * k, H - k, p)
        a = corr()
        if not np.array_equal(a, ref):
            bad += 1
            d = np.nonzero(a != ref)[0]
            err = np.abs(a[d].astype(np.float64) - ref[d])
            print("iter %d: %d scores differ, idx %d..%d, max |diff| %.3e at %d" %
                  (it, d.size, d[0], d[-1], err.max(), d[int(np.argmax(err))]), flush=True)
            hop = 1655808
            blocks = np.unique(d // hop)
            print("   blocks touched:", blocks[:20], "n_blocks", blocks.size, flush=True)
        if it % 20 == 0:
            print("iter", it, "bad so far", bad, flush=True)
    print("done: %d iterations, %d mismatches" % (iters, bad))


if __name__ == "__main__":
    main()
