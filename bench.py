#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X audio matcher.

Workload (BASELINE.json configs[1]): one 10 s mono 44.1 kHz f32 needle matched
against 1 h haystacks that are already resident in HBM.  One "step" = one pass
of the whole hot path (overlap-save correlation, score scan, peak pick,
cross-chunk merge = calc_chunks, audio_matcher.rs:88-141) over one haystack per
rank.  Haystacks are independent, so ranks shard them with no data-path
collective (weak scaling); torch.distributed (gloo) is used only for the
barrier and the max-over-ranks of the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
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


def plant_offsets(k: int):
    """t_m = 600*sr*m + 30*sr + 17*k + 1234, m = 0..5."""
    return [600 * SR * m + 30 * SR + 17 * k + 1234 for m in range(6)]


def make_inputs(am, device: int, n_hay: int, rank: int):
    s = NEEDLE_S * SR
    h = HAY_S * SR
    needle = am.synth_uniform_device(device, s, seed=1, stream=0)
    algo = am.HipConvolve.from_device(device, needle.ptr, s)
    hays = []
    for i in range(n_hay):
        k = rank * n_hay + i
        buf = am.synth_uniform_device(device, h, seed=1, stream=k + 1)
        for t in plant_offsets(k):
            am.axpy_device(device, buf, t, needle.ptr, s, 1.0)
        hays.append((k, buf))
    return needle, algo, hays


def cpu_baseline(n_chunks: int, threads: int):
    """The oracle (a port: the reference is Rust and cannot be built here) run the
    way the reference runs: per 60 s chunk a window of chunk+overlap samples,
    transforms of the non-power-of-two length w+s-1 in f32, needle re-transformed
    per chunk, chunks fanned out over `threads` host threads (rayon par_bridge,
    audio_matcher.rs:114)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    s = NEEDLE_S * SR
    chunk = CHUNK_S * SR
    h = n_chunks * chunk
    needle = po.synth_uniform(1, 0, 0, s)
    hay = po.synth_uniform(1, 1, 0, h)
    for t in plant_offsets(0):
        if t + s <= h:
            hay[t:t + s] += needle
    t0 = time.perf_counter()
    peaks = po.calc_chunks(SR, hay, needle, chunk, s, 0.13, 480 * SR, 480.0, scale=po.SCALE_LIB,
                           fft=po.FFT_REFERENCE, prec=po.PREC_F32, threads=threads)
    dt = time.perf_counter() - t0
    expect = [t for t in plant_offsets(0) if t + s <= h]
    ok = [p[0] for p in peaks] == expect
    return {"value": h / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n_chunks} x 60 s chunks of the 1 h haystack ({h} samples), window {chunk + s}, "
                      f"pad {chunk + 2 * s - 1} (Bluestein, f32), {dt:.2f} s wall, offsets_ok={ok}"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-steps", type=int, default=100,
                    help="untimed steps run during setup, before the W warmup steps, so that the GPU has left its "
                         "idle power state (the first ~50 ms of load run at lower clocks)")
    ap.add_argument("--haystacks-per-rank", type=int, default=2,
                    help="distinct resident 1 h haystacks each rank cycles through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the CPU sample (0 = three per thread, at most the whole hour)")
    ap.add_argument("--log-n", type=int, default=0)
    ap.add_argument("--pairs-per-group", type=int, default=0)
    ap.add_argument("--k2-variant", type=int, default=-1)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (experiments)")
    ap.add_argument("--half-pipeline", action="store_true",
                    help="BASELINE config 5 precision: half-precision storage of the work matrix (not the headline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import build as am_build
    if rank == 0:
        am_build.build_library()
    if dist is not None:
        dist.barrier()
    import audiomatch_amd as am
    ndev = am.device_count()
    if ndev < 1:
        raise RuntimeError("bench.py needs a HIP device; there is no CPU fallback")
    device = local_rank % ndev
    if args.log_n:
        am.set_option("log_n", args.log_n)
    if args.pairs_per_group:
        am.set_option("pairs_per_group", args.pairs_per_group)
    if args.k2_variant >= 0:
        am.set_option("k2_variant", args.k2_variant)
    if args.half_pipeline:
        am.set_option("half_pipeline", 1)
    if args.lanes:
        am.set_option("lanes", args.lanes)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        am.set_option(k_, int(v_))

    needle, algo, hays = make_inputs(am, device, args.haystacks_per_rank, rank)
    cfg = am.Config(chunk_size_s=CHUNK_S, overlap_length_s=NEEDLE_S, distance_s=480.0, prominence=0.13)
    params = cfg.params(SR, am.Scale.LIB)
    h = HAY_S * SR

    def step(i):
        k, buf = hays[i % len(hays)]
        return k, algo.match_device(buf.ptr, h, params, cap=64)

    def sync():
        am._check(am.lib().am_device_synchronize(device))
        if dist is not None:
            dist.barrier()

    for i in range(args.ramp_steps):
        step(i)
    for i in range(args.warmup):
        k, peaks = step(i)
        assert [p.start for p in peaks] == plant_offsets(k), (k, peaks)

    # Timed region: HIP events bracket only the dominant kernel (k2_rows) so that the
    # per-launch duration of the roofline is measured live without loading every
    # launch with event records; the full per-kernel breakdown comes from a short
    # untimed pass afterwards.
    KN = ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks")
    am.set_option("profile_mask", 1 << KN.index("k2_rows"))
    sync()
    with am.Profile(device) as prof:
        t0 = time.perf_counter()
        results = [step(i) for i in range(args.steps)]
        am._check(am.lib().am_device_synchronize(device))
        local_dt = time.perf_counter() - t0
        dom_timed = prof.query("k2_rows")
    am.set_option("profile_mask", -1)
    extra_steps = min(args.steps, 5)
    with am.Profile(device) as prof:
        for i in range(extra_steps):
            step(i)
        kern = {name: (prof.query(name)[0] * args.steps / extra_steps, prof.query(name)[1] * args.steps // extra_steps)
                for name in KN}
    kern["k2_rows"] = dom_timed
    if dist is not None:
        import torch
        dist.barrier()
        t = torch.tensor([local_dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        dt = local_dt
    for k, peaks in results:
        assert [p.start for p in peaks] == plant_offsets(k), (k, peaks)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_samples = float(h) * args.steps * world
    value = total_samples / dt
    # ---- roofline of the dominant kernel (algorithmic bytes) ----
    log_n = am.get_option("log_n") or 0
    s = NEEDLE_S * SR
    if not log_n:
        log_n = 21   # the library's plan for every large problem (am_api.hip pick_log_n)
    n_fft = 2 ** log_n
    hop = n_fft - s + 1
    if hop >= 8192:
        hop = hop // 1024 * 1024
    out_count = h - s + 1
    nblocks = -(-out_count // hop)
    npairs = (nblocks + 1) // 2
    # algorithmic bytes per step and kernel (DESIGN.md "Kernels"): every launch
    # covers all pairs of one haystack
    per_step_bytes = {
        "k1_cols_fwd": npairs * n_fft * (8 + 8),          # two f32 blocks in, complex out
        # complex in, complex out; the needle spectrum (8 B per point of ONE transform) is
        # shared by all pairs and has to come from HBM once per launch, not once per pair
        "k2_rows": npairs * n_fft * (8 + 8) + n_fft * 8,
        "k3_cols_inv": npairs * n_fft * 8 + (out_count // 32) * 8,  # complex in, (min,max) per 32 scores out
    }
    dom = "k2_rows"   # the dominant kernel by time (checked below against the untimed breakdown)
    dom_ms, dom_launches = kern[dom]
    dom_bytes_per_launch = per_step_bytes[dom] * args.steps / max(dom_launches, 1)
    dom_avg_s = dom_ms * 1e-3 / max(dom_launches, 1)
    achieved = dom_bytes_per_launch / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    kernel_ms_total = sum(v[0] for v in kern.values())
    # whole-pipeline figure from the wall clock of the timed region (local rank)
    pipe_gbs = (SURVEY_BYTES_PER_SAMPLE * float(h) * args.steps) / local_dt / 1e9
    # what this design has to move per step (K1 + K2 + K3 above), and the same from the PMC counters
    design_bytes = float(sum(per_step_bytes.values()))
    pmc_bytes = [pmc_traffic(k_) for k_ in ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks")]
    pmc_total = float(sum(pmc_bytes)) if all(b is not None for b in pmc_bytes) else None
    out = {
        "metric": METRIC, "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (work matrix stored as f16)" if args.half_pipeline else "f32",
        "data": "synthetic",
        "config": {"workload": "1 x 10 s mono 44.1 kHz f32 needle vs 1 x 1 h haystack per rank per step, "
                               "resident in HBM (BASELINE configs[1]); 6 planted hits per haystack",
                   "needle_samples": s, "haystack_samples": h, "fft_log2": log_n, "hop": hop,
                   "ramp_steps": args.ramp_steps,
                   "sharding": f"{world} rank(s), independent haystacks, no collective"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom),
                     "bytes_per_launch": dom_bytes_per_launch, "avg_launch_us": dom_avg_s * 1e6,
                     "launches": dom_launches},
        "roofline_pipeline": {"bound": "hbm", "achieved": pipe_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": pipe_gbs / HBM_PEAK_GBS,
                              "bytes_per_sample": SURVEY_BYTES_PER_SAMPLE,
                              "basis": "SURVEY.md 8(d) model bytes (28 N per block at N = 2^22) over the wall clock of the timed region",
                              "design_bytes_per_step": design_bytes,
                              "design_frac": design_bytes * args.steps / local_dt / 1e9 / HBM_PEAK_GBS,
                              "pmc_bytes_per_step": pmc_total,
                              "pmc_frac": (pmc_total * args.steps / local_dt / 1e9 / HBM_PEAK_GBS) if pmc_total else None,
                              "dominant_by_time": max(KN, key=lambda n_: kern[n_][0]),
                              "kernel_ms_per_step": {n_: v[0] / args.steps for n_, v in kern.items()}},
    }
    if world == 1:
        # the boundary also accepts host buffers (am_match): report the PCIe-inclusive
        # rate next to the resident one (never used as `value`)
        k0, buf0 = hays[0]
        host = buf0.to_numpy("float32", h)
        algo.match(host, params)
        t0 = time.perf_counter()
        pk = algo.match(host, params)
        te = time.perf_counter() - t0
        assert [p.start for p in pk] == plant_offsets(k0)
        out["end_to_end_host_buffer"] = {"value": h / te, "unit": "samples/s",
                                         "note": "am_match from pageable host memory: H2D copy + match, 1 haystack"}
        del host
    if world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 16)
        chunks = args.cpu_chunks or min(3 * threads, HAY_S // CHUNK_S)
        out["cpu_baseline"] = cpu_baseline(chunks, threads)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
