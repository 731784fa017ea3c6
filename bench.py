#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X audio matcher.

Workload (BASELINE.json configs[1] shape): one 10 s mono 44.1 kHz f32 needle matched
against 1 h haystacks that are already resident in HBM.  One "step" = one pass of the
whole hot path (overlap-save correlation, score scan, peak pick, cross-chunk merge =
calc_chunks, audio_matcher.rs:88-141) over one batch of `--haystacks-per-step` distinct
haystacks per rank through the batch entry point (am_match_batch_device = the per-file
loop of matcher::run, matcher/mod.rs:42-87).

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under torch.distributed.run the ranks come from the
environment; without it the parent process -- before it makes any HIP call -- starts N
fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and relays rank
0's JSON line.  Haystacks are independent, so ranks shard them with no data-path
collective (no RCCL traffic); torch.distributed (gloo) carries only the barriers and
the max-over-ranks of the timed region.

Default mode: weak scaling (fixed work per rank).  Every run also measures the
north-star batch -- 1000 haystacks sharded k mod N over the ranks, strong scaling --
once and reports it as "batch_1000" (never as `value`); `--total-haystacks T` makes
that strong-scaling batch the timed step itself.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "audio-matcher_amd", "python"), os.path.join(ROOT, "audio-matcher_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SR = 44100
NEEDLE_S = 10
HAY_S = 3600
CHUNK_S = 60
SURVEY_BYTES_PER_SAMPLE = 31.29      # 28*N per block of N-S+1 samples at N = 2^22
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
METRIC = "audio samples/s matched (whole node), 10 s needle vs 1 h haystack, 1/2/4/8 GPU"
RESIDENT_BUDGET_BYTES = 200e9        # haystacks kept resident per GPU in one wave (of 288 GB)
KN = ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks")


def plant_offsets(k: int):
    """t_m = 600*sr*m + 30*sr + 17*k + 1234, m = 0..5."""
    return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]


def shard(n_items: int, rank: int, world: int):
    """Haystack k -> rank k mod world (the C ABI's am_shard_plan)."""
    from audiomatch_amd import sharding
    return sharding.shard_indices(n_items, rank, world)


def fill_haystack(am, device, buf, k, needle_ptr, s, h):
    am._check(am.lib().am_synth_uniform_device(device, buf.ptr, 1, k + 1, 0, h, 0.25))
    for t in plant_offsets(k):
        am.axpy_device(device, buf, t, needle_ptr, s, 1.0)


def cpu_baseline(n_chunks: int, threads: int, policy: str):
    """The oracle (a port: the reference is Rust and cannot be built here) on the GPU box's
    host cores.  policy "reference": as the reference runs -- per 60 s
    chunk a window of chunk+overlap samples, transforms of the non-power-of-two length
    w+s-1 in f32, needle re-transformed per chunk, chunks fanned out over `threads` host
    threads (rayon par_bridge, audio_matcher.rs:114).  policy "pow2_cached": row C1, the
    fair CPU implementation -- power-of-two padding, needle spectrum computed once."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    s = NEEDLE_S * SR
    chunk = CHUNK_S * SR
    h = n_chunks * chunk
    needle = po.synth_uniform(1, 0, 0, s)
    hay = po.synth_uniform(1, 1, 0, h)
    plants = [t for t in plant_offsets(0) if t + s <= h] if n_chunks > 1 else [20 * SR]
    for t in plants:
        hay[t:t + s] += needle
    fft = po.FFT_REFERENCE if policy == "reference" else po.FFT_POW2_CACHED
    t0 = time.perf_counter()
    peaks = po.calc_chunks(SR, hay, needle, chunk, s, 0.13, 480 * SR, 480.0, scale=po.SCALE_LIB,
                           fft=fft, prec=po.PREC_F32, threads=threads)
    dt = time.perf_counter() - t0
    ok = [p[0] for p in peaks] == plants
    pad = chunk + 2 * s - 1
    how = f"pad {pad} (Bluestein, f32), needle re-transformed per chunk" if policy == "reference" \
        else f"pad {1 << (pad - 1).bit_length()} (power of two, f32), needle spectrum cached"
    return {"value": h / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n_chunks} x 60 s chunk(s) ({h} samples), window {chunk + s if n_chunks > 1 else chunk}, {how}, "
                      f"{dt:.2f} s wall, offsets_ok={ok}"}


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic.json, written by tools/pmc_summary.py from separate
    FETCH_SIZE / WRITE_SIZE passes of this same command, with the gfx950
    corrections of MI355X_MICROARCH.md); None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path))["kernels"][kernel]["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ramp-steps", type=int, default=12,
                    help="untimed steps run during setup, before the W warmup steps, so that the GPU has left its "
                         "idle power state (the first ~50 ms of load run at lower clocks)")
    ap.add_argument("--haystacks-per-step", type=int, default=8,
                    help="distinct resident 1 h haystacks each rank matches per step (one am_match_batch_device call)")
    ap.add_argument("--total-haystacks", type=int, default=0,
                    help="strong scaling: a step is this many haystacks in total, haystack k on rank k mod N "
                         "(the north-star batch is 1000)")
    ap.add_argument("--no-batch-1000", action="store_true", help="skip the extra 1000-haystack strong-scaling leg")
    ap.add_argument("--dry-shard", action="store_true",
                    help="print every rank's shard of --total-haystacks (default 1000) and stop before any GPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the dense / tonal / host-buffer side measurements")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the CPU sample (0 = three per thread, at most the whole hour)")
    ap.add_argument("--log-n", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (experiments)")
    ap.add_argument("--half-pipeline", type=int, nargs="?", const=1, default=0,
                    help="BASELINE config 5 precision (not the headline): 1 = work matrix stored as f16, "
                         "2 = K2's butterflies in packed f16 as well")
    return ap.parse_args()


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args) -> int:
    """--gpus N without a launcher: this process has made no HIP call (it only compiles);
    it starts N fresh children, one per GPU, and relays rank 0's line."""
    import build as am_build
    am_build.build_library()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(args.gpus))
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=e,
                                      stdout=subprocess.PIPE, text=True))
    rc = 0
    for r, p in enumerate(procs):
        out, _ = p.communicate()
        if out:
            sys.stdout.write(out)
        rc = max(rc, p.returncode if p.returncode >= 0 else 1)
    sys.stdout.flush()
    return rc


class Rank:
    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch one rank per GPU "
                             f"(python bench.py --gpus N, or torch.distributed.run --nproc-per-node N bench.py --gpus N)")

    def init_dist(self):
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_all(self, v: float) -> float:
        if self.dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        if self.dist is None:
            return [obj]
        parts = [None] * self.world
        self.dist.all_gather_object(parts, obj)
        return parts

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def strong_batch(am, R, device, algo, needle, params, total, steps, s, h, ranks_per_device=1):
    """`steps` passes over a batch of `total` haystacks, haystack k on rank k mod world.
    Resident waves: what fits the per-GPU budget is generated on device (untimed), matched
    with one am_match_batch_device call (timed), verified, and its buffers reused for the
    next wave.  Returns (seconds of the slowest rank summed over waves and steps, wave size)."""
    mine = shard(total, R.rank, R.world)
    per_wave = max(1, int(RESIDENT_BUDGET_BYTES // ranks_per_device // (4 * h)))
    n_buf = min(per_wave, max(1, len(mine)))
    bufs = [am.DeviceBuffer(device, 4 * h) for _ in range(n_buf)]
    local = 0.0
    resident = None
    # every rank goes through the same number of waves (an empty one still takes part in the barrier)
    n_waves = max(1, int(R.max_all(float((len(mine) + n_buf - 1) // n_buf))))
    for _ in range(steps):
        for wi in range(n_waves):
            wave = mine[wi * n_buf:(wi + 1) * n_buf]
            if wave and wave != resident:             # a shard that fits one wave is generated once
                for b, k in zip(bufs, wave):
                    fill_haystack(am, device, b, k, needle.ptr, s, h)
                resident = wave
            am._check(am.lib().am_device_synchronize(device))
            R.barrier()
            t0 = time.perf_counter()
            res = algo.match_batch_device([b.ptr for b in bufs[:len(wave)]], [h] * len(wave), params, cap_per_hay=16) if wave else []
            am._check(am.lib().am_device_synchronize(device))
            local += time.perf_counter() - t0
            for k, peaks in zip(wave, res):
                assert [p.start for p in peaks] == plant_offsets(k), (k, peaks)
    for b in bufs:
        b.free()
    R.barrier()
    return local, n_buf


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    R = Rank(args)
    # stdout carries exactly one JSON line (rank 0's); everything else a library may print there
    # (gloo's connection messages, for one) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    if args.dry_shard:
        total = args.total_haystacks or 1000
        mine = shard(total, R.rank, R.world)
        emit(json.dumps({"rank": R.rank, "world": R.world, "total_haystacks": total, "count": len(mine),
                         "first": mine[:3], "last": mine[-1] if mine else None}))
        import audiomatch_amd as am
        if am.device_count() < 1:
            raise SystemExit(f"rank {R.rank}: bench.py needs a HIP device; there is no CPU fallback")
        return

    R.init_dist()
    import build as am_build
    if R.rank == 0:
        am_build.build_library()
    R.barrier()
    import audiomatch_amd as am
    ndev = am.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py needs a HIP device; there is no CPU fallback")
    device = R.local_rank % ndev
    if args.log_n:
        am.set_option("log_n", args.log_n)
    if args.half_pipeline:
        am.set_option("half_pipeline", args.half_pipeline)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        am.set_option(k_, int(v_))

    s, h = NEEDLE_S * SR, HAY_S * SR
    needle = am.synth_uniform_device(device, s, seed=1, stream=0)
    algo = am.HipConvolve.from_device(device, needle.ptr, s)
    cfg = am.Config(chunk_size_s=CHUNK_S, overlap_length_s=NEEDLE_S, distance_s=480.0, prominence=0.13)
    params = cfg.params(SR, am.Scale.LIB)
    strong = args.total_haystacks > 0
    B = args.haystacks_per_step
    rpd = -(-R.world // ndev)          # ranks sharing one device (1 on a full node; a rehearsal on fewer GPUs shares)

    def sync():
        am._check(am.lib().am_device_synchronize(device))

    hays = []
    if not strong:
        for i in range(B):
            k = R.rank * B + i
            buf = am.DeviceBuffer(device, 4 * h)
            fill_haystack(am, device, buf, k, needle.ptr, s, h)
            hays.append((k, buf))
        ptrs, lens = [b.ptr for _, b in hays], [h] * B

        def step():
            return algo.match_batch_device(ptrs, lens, params, cap_per_hay=16)

        def check(res):
            for (k, _), peaks in zip(hays, res):
                assert [p.start for p in peaks] == plant_offsets(k), (k, peaks)

        for _ in range(args.ramp_steps):
            step()
        for _ in range(args.warmup):
            check(step())
        # Timed region: HIP events bracket only the dominant kernel (k2_rows), on the stream it
        # is launched on, so that its per-launch duration is measured live without loading every
        # launch with event records; the per-kernel breakdown comes from a short untimed pass.
        am.set_option("profile_mask", 1 << KN.index("k2_rows"))
        sync()
        R.barrier()
        with am.Profile(device) as prof:
            t0 = time.perf_counter()
            results = [step() for _ in range(args.steps)]
            sync()
            local_dt = time.perf_counter() - t0
            dom_timed = prof.query("k2_rows")
        R.barrier()
        for res in results:
            check(res)
        units_per_step = float(h) * B * R.world
        launches_per_step = B
    else:
        am.set_option("profile_mask", 1 << KN.index("k2_rows"))
        strong_batch(am, R, device, algo, needle, params, args.total_haystacks, max(1, min(args.warmup, 1)), s, h, rpd)
        with am.Profile(device) as prof:
            local_dt, wave = strong_batch(am, R, device, algo, needle, params, args.total_haystacks, args.steps, s, h, rpd)
            dom_timed = prof.query("k2_rows")
        units_per_step = float(h) * args.total_haystacks
        launches_per_step = len(shard(args.total_haystacks, R.rank, R.world))
        # a resident set for the untimed per-kernel breakdown below
        buf = am.DeviceBuffer(device, 4 * h)
        fill_haystack(am, device, buf, R.rank, needle.ptr, s, h)
        hays = [(R.rank, buf)]
        ptrs, lens = [buf.ptr], [h]

        def step():
            return algo.match_batch_device(ptrs, lens, params, cap_per_hay=16)

    dt = R.max_all(local_dt)
    per_rank = R.gather(round(local_dt, 6))

    # untimed: per-kernel breakdown of a few steps (every launch bracketed by events)
    am.set_option("profile_mask", -1)
    extra_steps = 3
    with am.Profile(device) as prof:
        for _ in range(extra_steps):
            step()
        kern = {name: (prof.query(name)[0] / extra_steps, prof.query(name)[1] // extra_steps) for name in KN}
    n_hay_step = len(ptrs)

    # the north-star batch as an extra leg: 1000 haystacks, strong scaling, one pass
    batch_1000 = None
    if not strong and not args.no_batch_1000:
        b_local, b_wave = strong_batch(am, R, device, algo, needle, params, 1000, 1, s, h, rpd)
        b_dt = R.max_all(b_local)
        b_parts = R.gather(round(b_local, 6))
        batch_1000 = {"value": 1000.0 * h / b_dt, "unit": "samples/s", "scaling": "strong", "n_gpus": R.world,
                      "haystacks": 1000, "seconds": b_dt, "per_rank_seconds": b_parts, "resident_wave": b_wave,
                      "note": "1000 x 1 h haystacks, haystack k on rank k mod N, generated on device in resident waves "
                              "(untimed), matched by am_match_batch_device (timed, max over ranks), all offsets verified"}

    if R.rank != 0:
        R.close()
        return

    value = units_per_step * args.steps / dt
    # ---- roofline of the dominant kernel (algorithmic bytes) ----
    log_n = am.get_option("log_n") or 0
    if not log_n:
        log_n = 21 if s <= 300000 else 22   # the library's plan for this needle (am_api.hip pick_log_n)
    n_fft = 2 ** log_n
    hop = n_fft - s + 1
    if hop >= 8192:
        hop = hop // 1024 * 1024
    out_count = h - s + 1
    nblocks = -(-out_count // hop)
    npairs = (nblocks + 1) // 2
    dense = am.get_option("dense_scores")
    per_hay_bytes = {
        "k1_cols_fwd": npairs * n_fft * (8 + 8),          # two f32 blocks in, complex out
        # complex in, complex out; the needle spectrum (8 B per point of ONE transform) is
        # shared by all pairs and has to come from HBM once per launch, not once per pair
        "k2_rows": npairs * n_fft * (8 + 8) + n_fft * 8,
        # complex in, (min,max) per 32 scores out; raw scores only when every tile is written
        "k3_cols_inv": npairs * n_fft * 8 + (out_count // 32) * 8 + (out_count * 4 if dense else 0),
    }
    dom = "k2_rows"
    dom_ms, dom_launches = dom_timed
    dom_avg_s = dom_ms * 1e-3 / max(dom_launches, 1)
    achieved = per_hay_bytes[dom] / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    hay_per_rank_step = launches_per_step
    pipe_gbs = (SURVEY_BYTES_PER_SAMPLE * float(h) * hay_per_rank_step * args.steps) / local_dt / 1e9
    design_bytes = float(sum(per_hay_bytes.values()))
    pmc_bytes = [pmc_traffic(k_) for k_ in KN]
    pmc_total = float(sum(pmc_bytes)) if all(b is not None for b in pmc_bytes) else None
    timed_ms = dt * 1e3
    out = {
        "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": R.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": {0: "f32", 1: "f32 (work matrix stored as f16)", 2: "f16 row butterflies, f16 work matrix, f32 column passes"}[min(args.half_pipeline, 2)],
        "data": "synthetic",
        "config": {"workload": (f"1 x 10 s mono 44.1 kHz f32 needle vs {args.total_haystacks} x 1 h haystacks per step in total, "
                                f"haystack k on rank k mod N, resident in HBM (BASELINE configs[2])" if strong else
                                f"1 x 10 s mono 44.1 kHz f32 needle vs {B} x 1 h haystacks per rank per step (one batch call), "
                                f"resident in HBM (BASELINE configs[1] shape); 6 planted hits per haystack, every result verified"),
                   "needle_samples": s, "haystack_samples": h, "haystacks_per_rank_per_step": hay_per_rank_step,
                   "fft_log2": log_n, "hop": hop, "ramp_steps": args.ramp_steps, "timed_region_ms": timed_ms,
                   "devices_visible_per_rank": ndev, "ranks_per_device": rpd, "per_rank_seconds": per_rank,
                   "sharding": f"{R.world} rank(s), independent haystacks, no collective"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom),
                     "bytes_per_launch": per_hay_bytes[dom], "avg_launch_us": dom_avg_s * 1e6,
                     "launches": dom_launches},
        "roofline_pipeline": {"bound": "hbm", "achieved": pipe_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": pipe_gbs / HBM_PEAK_GBS,
                              "bytes_per_sample": SURVEY_BYTES_PER_SAMPLE,
                              "basis": "SURVEY.md 8(d) model bytes (28 N per block at N = 2^22) over the wall clock of the timed region",
                              "design_bytes_per_haystack": design_bytes,
                              "design_frac": design_bytes * hay_per_rank_step * args.steps / local_dt / 1e9 / HBM_PEAK_GBS,
                              "pmc_bytes_per_haystack": pmc_total,
                              "pmc_frac": (pmc_total * hay_per_rank_step * args.steps / local_dt / 1e9 / HBM_PEAK_GBS) if pmc_total else None,
                              "dominant_by_time": max(KN, key=lambda n_: kern[n_][0]),
                              "kernel_ms_per_haystack": {n_: v[0] / n_hay_step for n_, v in kern.items()},
                              "kernel_bytes_per_haystack": per_hay_bytes},
    }
    if timed_ms < 100.0:
        out["config"]["warning"] = f"timed region {timed_ms:.1f} ms < 100 ms: raise --steps or --haystacks-per-step"
    if batch_1000 is not None:
        out["batch_1000"] = batch_1000
    if R.world == 1 and not strong and not args.no_extra_legs:
        out["side_measurements"] = side_measurements(am, device, algo, needle, params, hays, s, h, args.steps)
    if R.world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 16)
        chunks = args.cpu_chunks or min(3 * threads, HAY_S // CHUNK_S)
        out["cpu_baseline"] = cpu_baseline(chunks, threads, "reference")
        out["cpu_baseline_pow2"] = cpu_baseline(chunks, threads, "pow2_cached")             # row C1
        out["cpu_baseline_config1"] = cpu_baseline(1, 1, "reference")                       # configs[0]: 10 s vs 60 s, one chunk
    emit(json.dumps(out))
    R.close()


def side_measurements(am, device, algo, needle, params, hays, s, h, steps):
    """Numbers that belong next to the headline (never `value`): the worst case of the
    sparse-score path, a signal whose scores are not white, and the host-buffer entry."""
    out = {}
    ptrs, lens = [b.ptr for _, b in hays], [h] * len(hays)

    def timed(fn, n):
        fn()
        am._check(am.lib().am_device_synchronize(device))
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        am._check(am.lib().am_device_synchronize(device))
        return (time.perf_counter() - t0) / n

    n = max(3, min(steps, 10))
    # (1) every raw score written by K3 (theta = -inf): +4 B per score of HBM writes
    am.set_option("dense_scores", 1)
    try:
        t = timed(lambda: algo.match_batch_device(ptrs, lens, params, cap_per_hay=16), n)
    finally:
        am.set_option("dense_scores", 0)
    out["dense_scores"] = {"value": h * len(ptrs) / t, "unit": "samples/s", "ms_per_haystack": t / len(ptrs) * 1e3,
                           "note": "option dense_scores=1: K3 writes all 158.3 M raw scores per haystack (+0.63 GB); identical results"}
    # (1b) BASELINE configs[4]'s precision on the same resident batch (opt-in, never the headline):
    # f16 work matrix, then packed-f16 butterflies as well; offsets checked, per-handle option
    ref = algo.match_batch_device(ptrs, lens, params, cap_per_hay=16)
    for level, name in ((1, "half_pipeline_level1"), (2, "half_pipeline_level2")):
        halg = am.HipConvolve.from_device(device, needle.ptr, s)
        halg.set_option("half_pipeline", level)
        for _ in range(6):
            halg.match_batch_device(ptrs, lens, params, cap_per_hay=16)
        t = timed(lambda: halg.match_batch_device(ptrs, lens, params, cap_per_hay=16), n)
        res = halg.match_batch_device(ptrs, lens, params, cap_per_hay=16)
        ok = all([p.start for p in r] == plant_offsets(k) for (k, _), r in zip(hays, res))
        worst = max([abs(p.height - q.height) for r, rr in zip(res, ref) for p, q in zip(r, rr)] or [float("nan")]) if ok else float("nan")
        out[name] = {"value": h * len(ptrs) / t, "unit": "samples/s", "ms_per_haystack": t / len(ptrs) * 1e3,
                     "offsets_ok": ok, "max_height_difference_to_f32": worst,
                     "note": ("option half_pipeline=1: work matrix stored as f16, f32 butterflies" if level == 1 else
                              "option half_pipeline=2: f16 work matrix and packed-f16 butterflies (K1, K2, K3's first pass)")}
        halg.close()
    # (1c) BASELINE configs[0] on the GPU: the one-chunk case the reference's own bench runs on the CPU
    # (10 s needle vs one 60 s haystack, plant at 20 s), resident, single calls -- latency, not throughput
    h0 = CHUNK_S * SR
    small = am.DeviceBuffer(device, 4 * h0)
    am._check(am.lib().am_synth_uniform_device(device, small.ptr, 1, 1, 0, h0, 0.25))
    am.axpy_device(device, small, 20 * SR, needle.ptr, s, 1.0)
    r0 = algo.match_device(small.ptr, h0, params)
    t = timed(lambda: algo.match_device(small.ptr, h0, params), 50)
    out["config0_60s_haystack_gpu"] = {"value": h0 / t, "unit": "samples/s", "ms_per_call": t * 1e3,
                                       "offsets_ok": [p.start for p in r0] == [20 * SR],
                                       "note": "configs[0] shape on the GPU (cpu_baseline_config1 is its CPU row): one 60 s haystack, "
                                               "one window, one block pair; a single synchronous call"}
    small.free()
    # (2) a signal whose scores are not white: slow drift + 440 Hz ripple (see make_tonal)
    tone_needle, tone_algo, tone_hay, plants = make_tonal(am, device, s, h)
    res = tone_algo.match_device(tone_hay.ptr, h, params)
    ok = [p.start for p in res] == plants
    with am.Profile(device) as prof:
        t = timed(lambda: tone_algo.match_device(tone_hay.ptr, h, params), max(2, n // 2))
        tk = {name: prof.query(name)[0] / max(prof.query(name)[1], 1) for name in KN}
    out["non_white_signal"] = {"value": h / t, "unit": "samples/s", "ms_per_haystack": t * 1e3, "offsets_ok": ok,
                               "n_peaks": len(res), "kernel_ms": tk,
                               "note": "needle with a DC offset and a 440 Hz tone, haystack with the tone and a 240 s drift: the score "
                                       "array drifts by +-0.13 with a +-0.03 ripple, so most tiles are written and thousands of ripple "
                                       "maxima per chunk need a prominence walk (none qualifies); exact results"}
    tone_hay.free()
    # (3) host buffers: pageable H2D copy + match (am_match), one haystack
    k0, buf0 = hays[0]
    host = buf0.to_numpy("float32", h)
    algo.match(host, params)
    t0 = time.perf_counter()
    pk = algo.match(host, params)
    te = time.perf_counter() - t0
    assert [p.start for p in pk] == plant_offsets(k0)
    out["end_to_end_host_buffer"] = {"value": h / te, "unit": "samples/s",
                                     "note": "am_match from pageable host memory: H2D copy + match, 1 haystack (PCIe-bound)"}
    # (4) the same through the pool (copy of haystack i+1 overlapped with the match of i)
    pool = am.Pool(needle.to_numpy("float32", s), [device])
    batch = [host, host, host, host]
    pool.match_batch(batch[:1], params)
    t0 = time.perf_counter()
    res = pool.match_batch(batch, params)
    te = time.perf_counter() - t0
    assert all([p.start for p in r] == plant_offsets(k0) for r in res)
    out["end_to_end_pool_host_buffers"] = {"value": len(batch) * h / te, "unit": "samples/s",
                                           "note": "am_pool_match_batch, 4 host haystacks, two-slot ring: copy overlapped with match"}
    pool.close()
    return out


def make_tonal(am, device, s, h):
    """A signal whose scores are not white (what speech or music against a jingle looks like):
    needle = noise + DC offset + 440 Hz tone; haystack = noise + the same tone + a slow drift
    (240 s period) + 6 planted needles.  The score array then follows the drift (amplitude
    ~0.13, monotone inside every 60 s chunk) with a 440 Hz ripple on top (amplitude ~0.03):
    far above the sparse-write threshold in most tiles, thousands of ripple maxima per chunk
    pass the necessary height test and need a (short) prominence walk, none of them
    qualifies.  Built on the host in f32 (one haystack) and uploaded."""
    import numpy as np
    rng = np.random.default_rng(5)
    t = np.arange(h, dtype=np.float64)
    tone = (0.0437 * np.sin(2 * np.pi * 440.0 / SR * t)).astype(np.float32)
    drift = (0.04 * np.sin(2 * np.pi * t / (240.0 * SR))).astype(np.float32)
    del t
    needle = rng.uniform(-0.25, 0.25, s).astype(np.float32) + tone[:s] + np.float32(0.1)
    hay = rng.uniform(-0.25, 0.25, h).astype(np.float32) + tone + drift
    del tone, drift
    plants = plant_offsets(0)
    for p0 in plants:
        hay[p0:p0 + s] += needle
    nbuf = am.DeviceBuffer.from_numpy(device, needle)
    algo = am.HipConvolve.from_device(device, nbuf.ptr, s)
    hbuf = am.DeviceBuffer.from_numpy(device, hay)
    return nbuf, algo, hbuf, plants


if __name__ == "__main__":
    main()
