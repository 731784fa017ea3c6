"""CPU test of the N > 1 path: two gloo ranks shard a haystack batch with no
data-path collective, gather the peak lists on the host and agree with the
single-process answer.  (On the GPU box the same helpers drive bench.py; here the
per-haystack matcher is the CPU oracle, which is what a rank would be checked
against.)"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def free_port() -> str:
    """A port nobody listens on right now (two test runs on one machine must not collide)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.path.join(%(root)r, "audio-matcher_amd", "python"))
    sys.path.insert(0, os.path.join(%(root)r, "oracle"))
    import torch.distributed as dist
    import pyoracle as po
    from audiomatch_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sr, s, n_hay = 4000, 4000, 5
    needle = po.synth_uniform(1, 0, 0, s)
    local = {}
    for k in sharding.shard_indices(n_hay, rank, world):
        hay = po.synth_uniform(1, k + 1, 0, 20 * sr)
        off = (3 + 2 * k) * sr + 17 * k
        hay[off:off + s] += needle
        pk = po.calc_chunks(sr, hay, needle, 8 * sr, s, 0.3, 480 * sr, 480.0)
        local[k] = [p[0] for p in pk]
    dist.barrier()
    merged = sharding.gather_results(local, dist)
    t = sharding.max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        print(json.dumps({"merged": {str(k): v for k, v in sorted(merged.items())}, "tmax": t}))
    dist.destroy_process_group()
""")


def test_two_rank_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    sr = 4000
    assert res["tmax"] == 2.0
    assert res["merged"] == {str(k): [(3 + 2 * k) * sr + 17 * k] for k in range(5)}


def test_shard_indices_partition():
    sys.path.insert(0, os.path.join(ROOT, "audio-matcher_amd", "python"))
    from audiomatch_amd import sharding
    for world in (1, 2, 3, 8):
        seen = sorted(k for r in range(world) for k in sharding.shard_indices(1000, r, world))
        assert seen == list(range(1000))
        assert all(sharding.owner_of(k, world) == r for r in range(world) for k in sharding.shard_indices(50, r, world))


BENCH_WORKER = textwrap.dedent("""
    import json, os, sys, types
    sys.path.insert(0, %(root)r)
    import bench
    R = bench.Rank(types.SimpleNamespace(gpus=2))
    R.init_dist()
    R.barrier()
    tmax = R.max_all(1.0 + R.rank)
    parts = R.gather({"rank": R.rank, "mine": bench.shard(10, R.rank, R.world)})
    if R.rank == 0:
        print(json.dumps({"tmax": tmax, "parts": parts}))
    R.close()
""")


def test_bench_control_plane_two_ranks(tmp_path):
    """bench.py's rank plumbing (gloo barrier, MAX of the timed region, gather of per-rank
    values, k mod N shard) with world_size 2 on the CPU -- the data path has no collective."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert res["tmax"] == 2.0
    assert res["parts"] == [{"rank": 0, "mine": [0, 2, 4, 6, 8]}, {"rank": 1, "mine": [1, 3, 5, 7, 9]}]
    # a world size that disagrees with --gpus is refused before anything else happens
    r = subprocess.run([sys.executable, "-c", "import sys, types; sys.path.insert(0, %r); import bench; bench.Rank(types.SimpleNamespace(gpus=4))" % ROOT],
                       env=dict(env, RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


LONG_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.path.join(%(root)r, "audio-matcher_amd", "python"))
    sys.path.insert(0, os.path.join(%(root)r, "oracle"))
    import numpy as np
    import torch.distributed as dist
    import pyoracle as po
    import audiomatch_amd as am
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sr, s = 2000, 2000
    chunk, overlap = 10 * sr, s
    needle = po.synth_uniform(3, 0, 0, s)
    n = 95 * sr + 321
    hay = po.synth_uniform(3, 1, 0, n)
    for t, g in ((4.0, 1.0), (19.5, 1.0), (21.0, 0.7), (49.5, 1.0), (50.2, 0.8), (93.0, 1.0)):
        off = int(t * sr)
        hay[off:off + s] += np.float32(g) * needle
    p = am.AmMatchParams(sr=sr, chunk=chunk, overlap=overlap, min_prominence=0.4, min_distance=sr // 2,
                         overshadow_distance_s=3.0, scale=1)
    w0, nw, a, cnt = am.long_plan(n, s, p, world, rank)        # a pure function: no device needed
    raw = []
    for i in range(w0, w0 + nw):                               # this rank's windows, as audio_matcher.rs:114-131 treats each
        win = hay[i * chunk:min(n, i * chunk + chunk + overlap)]
        sc = po.correlate(win, needle, po.MODE_VALID, po.SCALE_LIB)
        raw += [(q[0] + i * chunk, q[1] + i * chunk, q[2], q[3]) for q in po.find_peaks(sc, 0.4, sr // 2)]
    parts = [None] * world
    dist.all_gather_object(parts, raw)
    if rank == 0:
        flat = [am.Peak(*q) for part in parts for q in part]
        merged = am.merge_peaks(p, flat)                       # host code of the library: one merge over the union
        whole = po.calc_chunks(sr, hay, needle, chunk, overlap, 0.4, sr // 2, 3.0)
        print(json.dumps({"merged": [q.start for q in merged], "whole": [q[0] for q in whole], "raw": len(flat),
                          "windows": [am.long_plan(n, s, p, world, r)[1] for r in range(world)]}))
    dist.destroy_process_group()
""")


def test_two_ranks_split_one_long_haystack(tmp_path):
    """ONE haystack over two ranks (second sentence; bench.py --long-haystack): am_long_plan gives each
    rank a window range, every rank picks its windows' peaks (the checker stands in for the device here), the lists are
    gathered on the host and the library's am_merge_peaks runs ONE sort + overshadow pass over the union -- the
    checker's answer for the whole haystack, including the weaker hit that is overshadowed across the cut."""
    script = tmp_path / "long_worker.py"
    script.write_text(LONG_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    sr = 2000
    assert res["windows"] == [5, 5]
    assert res["merged"] == res["whole"] == [int(t * sr) for t in (4.0, 19.5, 49.5, 93.0)]
    assert res["raw"] > len(res["merged"])
