#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X audio matcher on BASELINE.json's configs.

  python bench.py --gpus N --steps K --warmup W [--config {2,3,4}]

--config 2 (default, BASELINE configs[1] shape per GPU / configs[2] with --total-haystacks 1000):
    one 10 s mono 44.1 kHz f32 needle against 1 h haystacks resident in HBM, f32 arithmetic;
    one step = am_match_batch_device over `--haystacks-per-step` distinct haystacks per rank.
--config 3 (BASELINE configs[3]): 32 needles against the same haystacks, the haystack's forward
    transform shared by the needles of a group (am_match_multi_batch_device); a unit is one haystack
    sample matched against one needle.
--config 4 (BASELINE configs[4]): 48 kHz interleaved i16 stereo frames, f16 butterflies
    (half_pipeline = 2), down-mix fused into the first kernel (am_match_pcm16_batch_device).

One "step" = one pass of the whole hot path (overlap-save correlation, score scan, peak pick,
cross-chunk merge = calc_chunks, audio_matcher.rs:88-141; the per-file loop of matcher::run,
matcher/mod.rs:42-87) over one batch per rank.

N > 1: one process per GPU.  Under torch.distributed.run the ranks come from the environment;
without it the parent process -- before it makes any HIP call -- starts N fresh child processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and relays rank 0's JSON line.  Haystacks are
independent, so ranks shard them with no data-path collective (no RCCL traffic);
torch.distributed (gloo) carries only the barriers and the max-over-ranks of the timed region.

Default mode: weak scaling (fixed work per rank).  With --config 2 every run also measures the
north-star batch -- 1000 haystacks sharded k mod N over the ranks, strong scaling -- once and
reports it as "batch_1000" (never as `value`); `--total-haystacks T` makes that strong-scaling
batch the timed step itself, for every config.

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
SURVEY_BYTES_PER_SAMPLE = 31.29      # 28*N per block of N-S+1 samples at N = 2^22 (one needle, f32)
SURVEY_BYTES_PER_NEEDLE_SAMPLE_32 = 18.44   # (16 + 16 K) N per block at K = 32 needles
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
METRIC = "audio samples/s matched (whole node), 10 s needle vs 1 h haystack, 1/2/4/8 GPU"
RESIDENT_BUDGET_BYTES = 200e9        # haystacks kept resident per GPU in one wave (of 288 GB)
KN = ("k1_cols_fwd", "k2_rows", "k3_cols_inv", "tile_stats", "peaks")


def plant_offsets(k: int, sr: int = SR):
    """t_m = 600*sr*m + 30*sr + 17*k + 1234, m = 0..5."""
    return [600 * sr * m + 30 * sr + 17 * k + 1234 for m in range(6)]


def shard(n_items: int, rank: int, world: int):
    """Haystack k -> rank k mod world (the C ABI's am_shard_plan)."""
    from audiomatch_amd import sharding
    return sharding.shard_indices(n_items, rank, world)


def plan_geometry(s: int, h: int, log_n: int = 0, tail_block: int = 0):
    """The library's block layout for a needle of s samples and a haystack of h (am_api.hip pick_log_n /
    plan_geometry / tail_plan): transform length, hop, blocks, pairs of the main pass and -- option tail_block, single-needle
    calls -- the transform length of the smaller plan an odd last block goes through (0: none)."""
    forced = bool(log_n)
    if not log_n:
        log_n = 21 if s <= 140000 else 22
        hop21 = (2 ** 21 - s + 1) // 1024 * 1024
        if log_n == 22 and 0 < h - s + 1 <= 2 * hop21:     # a short haystack: one pair of 2^21 blocks is enough
            log_n = 21
    n_fft = 2 ** log_n
    hop = n_fft - s + 1
    if hop >= 8192:
        hop = hop // 1024 * 1024
    out_count = h - s + 1
    nblocks = -(-out_count // hop)
    npairs, tail_n_fft = (nblocks + 1) // 2, 0
    if tail_block and not forced and log_n >= 22 and nblocks % 2 == 1 and nblocks >= 3 and hop % 1024 == 0:
        rest = out_count - (nblocks - 1) * hop
        for lt in range(21, log_n):
            hop_t = (2 ** lt - s + 1) // 1024 * 1024
            if hop_t >= 8192 and 2 * hop_t >= rest:
                npairs, tail_n_fft = npairs - 1, 2 ** lt
                break
    return {"log_n": log_n, "n_fft": n_fft, "hop": hop, "out_count": out_count, "nblocks": nblocks, "npairs": npairs,
            "tail_n_fft": tail_n_fft}


# ---------------------------------------------------------------------------
# The three workloads.  Each knows how to make its needle(s), fill a resident haystack, run one
# batch through the library, verify a result, and what its kernels move through HBM by design.
# ---------------------------------------------------------------------------
class Workload:
    config = 2
    sr = SR
    n_needles = 1
    default_batch = 8
    dominant = "k2_rows"
    dtype = "f32"
    tail_block = True      # the engine of this workload sends a haystack's odd last block through the smaller plan

    def __init__(self, am, device, args):
        self.am, self.device, self.args = am, device, args
        self.s, self.h = NEEDLE_S * self.sr, HAY_S * self.sr
        self.hay_bytes = 4 * self.h          # an f32 sample and an i16 stereo frame are both 4 bytes
        cfg = am.Config(chunk_size_s=CHUNK_S, overlap_length_s=NEEDLE_S, distance_s=480.0, prominence=0.13)
        self.params = cfg.params(self.sr, am.Scale.LIB)
        self.geo = plan_geometry(self.s, self.h, am.get_option("log_n") or 0, am.get_option("tail_block") if self.tail_block else 0)

    # -- config 2: one f32 needle ---------------------------------------------------------------
    def setup(self):
        am = self.am
        self.needle = am.synth_uniform_device(self.device, self.s, seed=1, stream=0)
        self.algo = am.HipConvolve.from_device(self.device, self.needle.ptr, self.s)

    def fill(self, buf, k):
        am = self.am
        am._check(am.lib().am_synth_uniform_device(self.device, buf.ptr, 1, k + 1, 0, self.h, 0.25))
        for t in plant_offsets(k, self.sr):
            am.axpy_device(self.device, buf, t, self.needle.ptr, self.s, 1.0)

    def match(self, ptrs):
        return self.algo.match_batch_device(ptrs, [self.h] * len(ptrs), self.params, cap_per_hay=16)

    def check(self, k, res):
        assert [p.start for p in res] == plant_offsets(k, self.sr), (k, res)

    def units_per_haystack(self):
        return float(self.h) * self.n_needles

    def kernel_bytes(self, dense=False, main_only=False):
        g = self.geo
        pts = g["npairs"] * g["n_fft"] + (0 if main_only else g["tail_n_fft"])
        # one point of the work matrix: a float2, or a half2 with --half-pipeline 1 / 2
        pt = 4 if self.am.get_option("half_pipeline") else 8
        spec = 4 if self.am.get_option("half_pipeline") >= 2 else 8   # (level 2 multiplies with an f16 copy of the spectrum)
        return {
            "k1_cols_fwd": pts * (8 + pt),                     # two f32 blocks in, one complex point out
            # complex in, complex out; the needle spectrum (one transform's worth) is shared by all
            # pairs through L2 and has to come from HBM once per launch, not once per pair
            "k2_rows": pts * (pt + pt) + g["n_fft"] * spec + (0 if main_only else g["tail_n_fft"] * spec),
            # complex in, (min,max) per 32 scores out; raw scores only where the pick can need them
            "k3_cols_inv": pts * pt + (g["out_count"] // 32) * 8 + (g["out_count"] * 4 if dense else 0),
        }

    def dominant_bytes_per_launch(self):
        # (the main pass's launch: the tail block's three launches are profiled apart, under "other" / tail_rows_*)
        return self.kernel_bytes(main_only=True)[self.dominant]

    def launches_per_haystack(self):
        return 1

    def describe(self, per_step, strong_total):
        if strong_total:
            return (f"1 x 10 s mono 44.1 kHz f32 needle vs {strong_total} x 1 h haystacks per step in total, haystack k on rank "
                    f"k mod N, resident in HBM (BASELINE configs[2])")
        return (f"1 x 10 s mono 44.1 kHz f32 needle vs {per_step} x 1 h haystacks per rank per step (one batch call), resident in "
                f"HBM (BASELINE configs[1] shape); 6 planted hits per haystack, every result verified")


class MultiNeedleWorkload(Workload):
    """BASELINE configs[3]: 32 needles (streams 2001..), each planted twice per haystack."""
    config = 3
    n_needles = 32
    default_batch = 2
    tail_block = True      # (needle groups of at least two: the tail's K1 once per haystack, its K2 and K3 once per group)

    def setup(self):
        am = self.am
        self.n_needles = self.args.needles
        self.group = am.get_option("needle_group")
        # (the library takes the tail block only when every needle group of a haystack runs the grouped K3)
        if self.geo["tail_n_fft"] and (self.group < 2 or self.n_needles < 2 or self.n_needles % self.group == 1 or not am.get_option("k3_group")):
            self.geo = plan_geometry(self.s, self.h, am.get_option("log_n") or 0, 0)
        self.needles = [am.synth_uniform_device(self.device, self.s, 1, 2001 + j) for j in range(self.n_needles)]
        self.algos = [am.HipConvolve.from_device(self.device, n.ptr, self.s) for n in self.needles]

    def plants(self, k, j):
        return [310 * self.sr + 1000 * j + 17 * k, 2010 * self.sr + 999 * j + 17 * k]

    def fill(self, buf, k):
        am = self.am
        am._check(am.lib().am_synth_uniform_device(self.device, buf.ptr, 1, k + 1, 0, self.h, 0.25))
        for j, n in enumerate(self.needles):
            for t in self.plants(k, j):
                am.axpy_device(self.device, buf, t, n.ptr, self.s, 1.0)

    def match(self, ptrs):
        return self.am.match_multi_batch_device(self.algos, ptrs, [self.h] * len(ptrs), self.params, cap_per_pair=8)

    def check(self, k, res):
        for j, r in enumerate(res):
            assert [p.start for p in r] == self.plants(k, j), (k, j, r)

    def kernel_bytes(self, dense=False, main_only=False):
        g = self.geo
        tail = 0 if main_only else g["tail_n_fft"]
        pts = g["npairs"] * g["n_fft"] + tail
        nn = self.n_needles
        ngroups = -(-nn // self.group)
        return {
            "k1_cols_fwd": pts * (8 + 8),                                        # once per haystack
            # per group launch: the haystack's rows read once, one inverse written per needle, every
            # needle's spectrum from HBM once (shared by all pairs through L2)
            "k2_rows": ngroups * pts * 8 + nn * (pts * 8 + g["n_fft"] * 8 + tail * 8),
            "k3_cols_inv": nn * (pts * 8 + (g["out_count"] // 32) * 8),
        }

    def dominant_bytes_per_launch(self):
        # (the main pass's group launches: a tail block's launches are profiled under "other")
        return self.kernel_bytes(main_only=True)["k2_rows"] / self.launches_per_haystack()

    def launches_per_haystack(self):
        return -(-self.n_needles // self.group)     # K2 group launches per haystack

    def describe(self, per_step, strong_total):
        what = f"{strong_total} x 1 h haystacks per step in total, haystack k on rank k mod N" if strong_total else \
            f"{per_step} x 1 h haystacks per rank per step (one am_match_multi_batch_device call)"
        return (f"{self.n_needles} x 10 s mono 44.1 kHz f32 needles vs {what}, resident in HBM (BASELINE configs[3]: haystack FFT "
                f"reused, needles in groups of {self.group}); a unit = one haystack sample against one needle; every needle planted "
                f"twice per haystack, all {2 * self.n_needles} offsets per haystack verified")


class Pcm16HalfWorkload(Workload):
    """BASELINE configs[4]: 48 kHz interleaved i16 stereo, half-precision butterflies."""
    config = 4
    sr = 48000
    dominant = "k1_cols_fwd"
    dtype = "f16 butterflies and f16 work matrix (K1, K2, first pass of K3), f32 score pass; i16 stereo input"

    def setup(self):
        am = self.am
        am.set_option("half_pipeline", self.args.half_pipeline or 2)
        self.level = am.get_option("half_pipeline")
        self.needle_pcm = am.synth_pcm16_stereo_device(self.device, self.s, seed=1, stream=0)
        self.algo = am.HipConvolve.from_pcm16(self.needle_pcm.to_numpy("int16", 2 * self.s), self.device)

    def fill(self, buf, k):
        am = self.am
        am._check(am.lib().am_synth_pcm16_stereo_device(self.device, buf.ptr, 1, k + 1, 0, self.h, 0.25))
        for t in plant_offsets(k, self.sr):
            am.add_pcm16_device(self.device, buf, t, self.needle_pcm.ptr, self.s)

    def match(self, ptrs):
        return self.algo.match_pcm16_batch_device(ptrs, [self.h] * len(ptrs), self.params, cap_per_hay=16)

    def kernel_bytes(self, dense=False, main_only=False):
        g = self.geo
        pts = g["npairs"] * g["n_fft"] + (0 if main_only else g["tail_n_fft"])
        return {
            "k1_cols_fwd": pts * (8 + 4),          # two i16 stereo frames in (4 B each), one half2 point out
            "k2_rows": pts * (4 + 4) + g["n_fft"] * 4 + (0 if main_only else g["tail_n_fft"] * 4),
            "k3_cols_inv": pts * 4 + (g["out_count"] // 32) * 8 + (g["out_count"] * 4 if dense else 0),
        }

    def describe(self, per_step, strong_total):
        what = f"{strong_total} x 1 h haystacks per step in total, haystack k on rank k mod N" if strong_total else \
            f"{per_step} x 1 h haystacks per rank per step (one am_match_pcm16_batch_device call)"
        return (f"1 x 10 s needle vs {what}: 48 kHz interleaved i16 stereo frames resident in HBM, down-mix fused into K1, "
                f"half_pipeline = {self.level} (BASELINE configs[4]); offsets verified (scores are within 1e-3 at this precision)")


WORKLOADS = {1: Workload, 2: Workload, 3: MultiNeedleWorkload, 4: Pcm16HalfWorkload}


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


def pmc_traffic(kernel: str, config: int = 2):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic.json for the headline, written by tools/pmc_summary.py from separate
    FETCH_SIZE / WRITE_SIZE passes of this same command, with the gfx950 corrections of
    MI355X_MICROARCH.md); None when no summary is committed for this config."""
    name = "pmc_traffic.json" if config in (1, 2) else f"pmc_traffic_config{config}.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        kernels = json.load(open(path))["kernels"]
    except (KeyError, ValueError):
        return None
    for name in (kernel, kernel + "_group"):       # (configs[3]: the row kernel of a needle group is summarised under its own name)
        if name in kernels:
            return kernels[name]["hbm_bytes_per_launch"]
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS),
                    help="BASELINE.json configs index: 2 = one needle (the headline; 1 is an alias), 3 = 32 needles, "
                         "4 = 48 kHz i16 stereo with f16 butterflies")
    ap.add_argument("--needles", type=int, default=32, help="--config 3: number of needles")
    ap.add_argument("--ramp-steps", type=int, default=12,
                    help="untimed steps run during setup, before the W warmup steps, so that the GPU has left its "
                         "idle power state (the first ~50 ms of load run at lower clocks)")
    ap.add_argument("--haystacks-per-step", type=int, default=0,
                    help="distinct resident 1 h haystacks each rank matches per step (one batch call; default 8, or 2 with --config 3)")
    ap.add_argument("--total-haystacks", type=int, default=0,
                    help="strong scaling: a step is this many haystacks in total, haystack k on rank k mod N "
                         "(the north-star batch is 1000)")
    ap.add_argument("--long-haystack", type=float, default=0.0, metavar="HOURS",
                    help="strong scaling on ONE haystack of this many hours (SURVEY.md 8e, second sentence): rank r matches the "
                         "window range am_long_plan gives it (am_match_part_device), rank 0 merges (am_merge_peaks); config 2 only")
    ap.add_argument("--no-batch-1000", action="store_true", help="skip the extra 1000-haystack strong-scaling leg")
    ap.add_argument("--dry-shard", action="store_true",
                    help="print every rank's shard of --total-haystacks (default 1000) and stop before any GPU work")
    ap.add_argument("--allow-shared-devices", action="store_true",
                    help="(accepted for older command lines) ranks that share a GPU are always allowed as a rehearsal: the line "
                         "then reports n_gpus = devices actually used and carries a warning")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the dense / non-white / host-buffer side measurements")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the CPU sample (0 = three per thread, at most the whole hour)")
    ap.add_argument("--profile-every", type=int, default=5,
                    help="the timed region brackets every n-th launch of the dominant kernel with HIP events (an event pair costs "
                         "the stream about 8 us per kernel boundary: bracketing all of them takes 2 %% off the headline)")
    ap.add_argument("--log-n", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (experiments)")
    ap.add_argument("--half-pipeline", type=int, nargs="?", const=1, default=0,
                    help="BASELINE configs[4] precision on the --config 2 workload (not the headline): 1 = work matrix stored as "
                         "f16, 2 = butterflies in packed f16 as well; with --config 4 the level to use (default 2)")
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


def strong_batch(am, R, W, total, steps, ranks_per_device=1):
    """`steps` passes over a batch of `total` haystacks, haystack k on rank k mod world.
    Resident waves: what fits the per-GPU budget is generated on device (untimed), matched
    with one batch call (timed), verified, and its buffers reused for the next wave.
    Returns (seconds of this rank summed over waves and steps, wave size)."""
    device = W.device
    mine = shard(total, R.rank, R.world)
    per_wave = max(1, int(RESIDENT_BUDGET_BYTES // ranks_per_device // W.hay_bytes))
    n_buf = min(per_wave, max(1, len(mine)))
    bufs = [am.DeviceBuffer(device, W.hay_bytes) for _ in range(n_buf)]
    local = 0.0
    resident = None
    # every rank goes through the same number of waves (an empty one still takes part in the barrier)
    n_waves = max(1, int(R.max_all(float((len(mine) + n_buf - 1) // n_buf))))
    for _ in range(steps):
        for wi in range(n_waves):
            wave = mine[wi * n_buf:(wi + 1) * n_buf]
            if wave and wave != resident:             # a shard that fits one wave is generated once
                for b, k in zip(bufs, wave):
                    W.fill(b, k)
                resident = wave
            am._check(am.lib().am_device_synchronize(device))
            R.barrier()
            t0 = time.perf_counter()
            res = W.match([b.ptr for b in bufs[:len(wave)]]) if wave else []
            am._check(am.lib().am_device_synchronize(device))
            local += time.perf_counter() - t0
            for k, r in zip(wave, res):
                W.check(k, r)
    for b in bufs:
        b.free()
    R.barrier()
    return local, n_buf


def long_haystack_leg(am, R, W, hours, steps, warmup):
    """ONE haystack of `hours` hours over all ranks (audio_matcher.rs:104-140 with the windows fanned out over GPUs
    instead of threads): rank r generates the samples of its part on its device (untimed), then per step: every rank
    matches its windows (am_match_part_device), the unmerged peaks are gathered on the host, rank 0 runs the one
    sort + overshadow pass (am_merge_peaks).  Returns (seconds per step: max over ranks incl. gather and merge,
    offsets_ok, part description)."""
    device = W.device
    sr, s = W.sr, W.s
    n = int(hours * 3600 * sr)
    w0, nw, a, cnt = am.long_plan(n, s, W.params, R.world, R.rank)
    plants = [600 * sr * m + 30 * sr + 1234 for m in range(int(hours * 6)) if 600 * sr * m + 30 * sr + 1234 + s <= n]
    buf = am.DeviceBuffer(device, 4 * max(cnt, 1))
    if cnt:
        am._check(am.lib().am_synth_uniform_device(device, buf.ptr, 1, 1, a, cnt, 0.25))
        for t in plants:                       # the part of every planted needle that falls into this rank's samples
            lo, hi = max(t, a), min(t + s, a + cnt)
            if lo < hi:
                am._check(am.lib().am_axpy_device(device, buf.ptr + 4 * (lo - a), W.needle.ptr + 4 * (lo - t), hi - lo, 1.0))

    def step():
        raw = am.match_part_device(W.algo, buf.ptr, cnt, W.params, nw, a, cap=4096) if nw else []
        parts = R.gather([(q.start, q.end, q.height, q.prominence) for q in raw])
        if R.rank != 0:
            return None
        flat = [am.Peak(*q) for part in parts for q in part]
        return am.merge_peaks(W.params, flat)

    for _ in range(max(1, warmup)):
        res = step()
    am._check(am.lib().am_device_synchronize(device))
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    local = time.perf_counter() - t0
    dt = R.max_all(local)
    ok = R.rank != 0 or [q.start for q in res] == plants
    desc = R.gather({"rank": R.rank, "windows": nw, "first_window": w0, "samples": cnt, "seconds": round(local, 6)})
    buf.free()
    return dt / steps, ok, desc, n, len(plants)


def device_identity(device: int) -> str:
    """Something that tells two physical GPUs apart across ranks: host name + the ordinal this rank
    uses (+ the visible-devices mask, which a launcher may set per rank)."""
    mask = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", ""))
    return f"{socket.gethostname()}:{mask}:{device}"


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
        emit(json.dumps({"rank": R.rank, "world": R.world, "config": args.config, "total_haystacks": total, "count": len(mine),
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
    # one rank per GPU is the contract: ranks that share a device are a rehearsal, and the line says so
    idents = R.gather(device_identity(device))
    devices_used = len(set(idents))
    rpd = max(idents.count(i) for i in set(idents))
    if rpd > 1 and R.rank == 0:
        print(f"bench.py: {R.world} ranks but only {devices_used} distinct GPU(s) ({ndev} visible per rank): one rank per GPU is the "
              f"contract -- this run is a rehearsal, its line reports n_gpus = {devices_used} and carries a warning", file=sys.stderr)
    if args.log_n:
        am.set_option("log_n", args.log_n)
    if args.half_pipeline and args.config != 4:
        am.set_option("half_pipeline", args.half_pipeline)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        am.set_option(k_, int(v_))

    W = WORKLOADS[args.config](am, device, args)
    W.setup()
    s, h = W.s, W.h
    strong = args.total_haystacks > 0
    B = args.haystacks_per_step or W.default_batch
    dom = W.dominant

    def sync():
        am._check(am.lib().am_device_synchronize(device))

    if args.long_haystack > 0:
        if args.config not in (1, 2):
            raise SystemExit("bench.py: --long-haystack runs the config 2 workload")
        am.set_option("profile_mask", 1 << KN.index(dom))
        with am.Profile(device) as prof:
            per_step, ok, parts, n_long, n_plants = long_haystack_leg(am, R, W, args.long_haystack, args.steps, args.warmup)
            dom_ms, dom_launches = prof.query(dom)
        if R.rank == 0:
            # rank 0's K2: its part's block pairs go through the row kernel in launches of at most `pairs_per_group`
            # pairs (each launch reads the needle spectrum once): bytes of all launches of a step over their time
            g = plan_geometry(s, parts[0]["samples"], am.get_option("log_n") or 0, am.get_option("tail_block")) \
                if parts[0]["samples"] >= s else None   # (npairs: the main pass's; a tail block is profiled under "other")
            ppg = am.get_option("pairs_per_group")
            launches_per_step = -(-g["npairs"] // ppg) if g else 1
            dom_bytes = ((g["npairs"] * g["n_fft"] * 16 + launches_per_step * g["n_fft"] * 8) / launches_per_step) if g else 0
            dom_avg_s = dom_ms * 1e-3 / max(dom_launches, 1)
            achieved = dom_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
            emit(json.dumps({
                "metric": METRIC, "value": n_long / per_step, "unit": "samples/s", "n_gpus": devices_used, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"1 x 10 s mono 44.1 kHz f32 needle vs ONE {args.long_haystack:g} h haystack split over the ranks by "
                                       f"window ranges (am_long_plan / am_match_part_device), one merge on rank 0 (am_merge_peaks); "
                                       f"{n_plants} planted hits, offsets verified: {ok}",
                           "baseline_config": 2, "haystack_samples": n_long, "ranks": R.world, "devices_used": devices_used,
                           "ranks_per_device": rpd, "parts": parts, "offsets_ok": ok,
                           "sharding": f"{R.world} rank(s), contiguous window ranges of one haystack, no collective (host gather of the peaks)"},
                "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": dom_bytes,
                             "avg_launch_us": dom_avg_s * 1e6, "launches": dom_launches,
                             "note": "rank 0's part; bytes_per_launch is the average over the launches of a step"}}))
            if not ok:
                raise SystemExit("bench.py: --long-haystack: the merged offsets differ from the planted ones")
        R.close()
        return

    hays = []
    if not strong:
        for i in range(B):
            k = R.rank * B + i
            buf = am.DeviceBuffer(device, W.hay_bytes)
            W.fill(buf, k)
            hays.append((k, buf))
        ptrs = [b.ptr for _, b in hays]

        def step():
            return W.match(ptrs)

        def check(res):
            for (k, _), r in zip(hays, res):
                W.check(k, r)

        for _ in range(args.ramp_steps):
            step()
        for _ in range(args.warmup):
            check(step())
        # Timed region: HIP events bracket only the dominant kernel, on the stream it is launched
        # on, and only every n-th of its launches (--profile-every; n = 5 walks through the positions
        # of an 8-haystack step), so that its per-launch duration is measured live without loading
        # the step with event records: a bracketed launch has 8 us of stream time on either side
        # where an unbracketed kernel boundary has 0 - 3.5 (profiles/r04/event_gaps.txt).  The
        # per-kernel breakdown comes from a short untimed pass.
        am.set_option("profile_mask", 1 << KN.index(dom))
        am.set_option("profile_every", max(1, args.profile_every))
        sync()
        R.barrier()
        with am.Profile(device) as prof:
            t0 = time.perf_counter()
            results = [step() for _ in range(args.steps)]
            sync()
            local_dt = time.perf_counter() - t0
            dom_timed = prof.query(dom)
        R.barrier()
        for res in results:
            check(res)
        hay_per_rank_step = B
        units_per_step = W.units_per_haystack() * B * R.world
    else:
        am.set_option("profile_mask", 1 << KN.index(dom))
        am.set_option("profile_every", max(1, args.profile_every))
        strong_batch(am, R, W, args.total_haystacks, max(1, min(args.warmup, 1)), rpd)
        with am.Profile(device) as prof:
            local_dt, wave = strong_batch(am, R, W, args.total_haystacks, args.steps, rpd)
            dom_timed = prof.query(dom)
        units_per_step = W.units_per_haystack() * args.total_haystacks
        hay_per_rank_step = len(shard(args.total_haystacks, R.rank, R.world))
        # a resident set for the untimed per-kernel breakdown below
        buf = am.DeviceBuffer(device, W.hay_bytes)
        W.fill(buf, R.rank)
        hays = [(R.rank, buf)]
        ptrs = [buf.ptr]

        def step():
            return W.match(ptrs)

    dt = R.max_all(local_dt)
    per_rank = R.gather(round(local_dt, 6))

    # untimed: per-kernel breakdown of a few steps (every launch bracketed by events)
    am.set_option("profile_mask", -1)
    am.set_option("profile_every", 1)
    extra_steps = 3
    with am.Profile(device) as prof:
        for _ in range(extra_steps):
            step()
        kern = {name: (prof.query(name)[0] / extra_steps, prof.query(name)[1] // extra_steps) for name in KN}
    n_hay_step = len(ptrs)

    # the north-star batch as an extra leg: 1000 haystacks, strong scaling, one pass
    batch_1000 = None
    if not strong and not args.no_batch_1000 and args.config in (1, 2):
        b_local, b_wave = strong_batch(am, R, W, 1000, 1, rpd)
        b_dt = R.max_all(b_local)
        b_parts = R.gather(round(b_local, 6))
        batch_1000 = {"value": 1000.0 * h / b_dt, "unit": "samples/s", "scaling": "strong", "n_gpus": devices_used,
                      "haystacks": 1000, "seconds": b_dt, "per_rank_seconds": b_parts, "resident_wave": b_wave,
                      "note": "1000 x 1 h haystacks, haystack k on rank k mod N, generated on device in resident waves "
                              "(untimed), matched by am_match_batch_device (timed, max over ranks), all offsets verified"}

    if R.rank != 0:
        R.close()
        return

    value = units_per_step * args.steps / dt
    # ---- roofline of the dominant kernel (algorithmic bytes) ----
    geo = W.geo
    dense = am.get_option("dense_scores")
    per_hay_bytes = W.kernel_bytes(bool(dense))
    dom_ms, dom_launches = dom_timed
    dom_avg_s = dom_ms * 1e-3 / max(dom_launches, 1)
    dom_bytes = W.dominant_bytes_per_launch()
    achieved = dom_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    design_bytes = float(sum(per_hay_bytes.values()))
    pmc_bytes = [pmc_traffic(k_, args.config) for k_ in KN]
    pmc_total = float(sum(pmc_bytes)) if all(b is not None for b in pmc_bytes) else None
    survey_bps = SURVEY_BYTES_PER_NEEDLE_SAMPLE_32 if args.config == 3 else SURVEY_BYTES_PER_SAMPLE
    survey_gbs = survey_bps * W.units_per_haystack() * hay_per_rank_step * args.steps / local_dt / 1e9
    design_gbs = design_bytes * hay_per_rank_step * args.steps / local_dt / 1e9
    timed_ms = dt * 1e3
    half = am.get_option("half_pipeline")
    out = {
        "metric": METRIC, "value": value, "unit": "samples/s" if W.n_needles == 1 else "needle-samples/s",
        "n_gpus": devices_used, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": W.dtype if args.config == 4 else
        {0: "f32", 1: "f32 (work matrix stored as f16)", 2: "f16 row butterflies, f16 work matrix, f32 column passes"}[min(half, 2)],
        "data": "synthetic",
        "config": {"workload": W.describe(B, args.total_haystacks if strong else 0),
                   "baseline_config": args.config, "needles": W.n_needles,
                   "needle_samples": s, "haystack_samples": h, "haystacks_per_rank_per_step": hay_per_rank_step,
                   "fft_log2": geo["log_n"], "hop": geo["hop"], "block_pairs_per_haystack": geo["npairs"],
                   "tail_block_fft_log2": (geo["tail_n_fft"].bit_length() - 1) if geo["tail_n_fft"] else None,
                   "ramp_steps": args.ramp_steps, "timed_region_ms": timed_ms,
                   "ranks": R.world, "devices_used": devices_used, "devices_visible_per_rank": ndev, "ranks_per_device": rpd,
                   "per_rank_seconds": per_rank,
                   "sharding": f"{R.world} rank(s), independent haystacks, no collective"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, args.config),
                     "bytes_per_launch": dom_bytes, "avg_launch_us": dom_avg_s * 1e6,
                     "launches": dom_launches,
                     "launch_sampling": f"HIP events around every {max(1, args.profile_every)}. launch of the kernel inside the timed region"},
        # the whole pipeline over the wall clock of the timed region, on the bytes THIS design moves through
        # HBM (K1 + K2 + K3 as in DESIGN.md section 5); the SURVEY.md 8(d) model (a two-pass forward and a
        # two-pass inverse transform per block) moves more bytes per sample, so its fraction is a model figure
        "roofline_pipeline": {"bound": "hbm", "achieved": design_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": design_gbs / HBM_PEAK_GBS,
                              "basis": "design bytes per haystack (K1 + K2 + K3 algorithmic bytes) over the wall clock of the timed region",
                              "design_bytes_per_haystack": design_bytes,
                              "survey_model_bytes_per_unit": survey_bps,
                              "survey_model_frac": survey_gbs / HBM_PEAK_GBS,
                              "pmc_bytes_per_haystack": pmc_total,
                              "pmc_frac": (pmc_total * hay_per_rank_step * args.steps / local_dt / 1e9 / HBM_PEAK_GBS) if pmc_total else None,
                              "dominant_by_time": max(KN, key=lambda n_: kern[n_][0]),
                              "kernel_ms_per_haystack": {n_: v[0] / n_hay_step for n_, v in kern.items()},
                              "kernel_bytes_per_haystack": per_hay_bytes},
    }
    if rpd > 1:
        out["config"]["warning_shared_devices"] = (f"{R.world} ranks shared {devices_used} GPU(s) (rehearsal): n_gpus counts "
                                                   f"devices, not ranks")
    if timed_ms < 100.0:
        out["config"]["warning"] = f"timed region {timed_ms:.1f} ms < 100 ms: raise --steps or --haystacks-per-step"
    if batch_1000 is not None:
        out["batch_1000"] = batch_1000
    if R.world == 1 and not strong and not args.no_extra_legs and args.config in (1, 2):
        out["side_measurements"] = side_measurements(am, device, W.algo, W.needle, W.params, hays, s, h, args.steps)
    if R.world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 16)
        chunks = args.cpu_chunks or min(3 * threads, HAY_S // CHUNK_S)
        out["cpu_baseline"] = cpu_baseline(chunks, threads, "reference")
        if args.config in (1, 2):
            out["cpu_baseline_pow2"] = cpu_baseline(chunks, threads, "pow2_cached")         # row C1
            out["cpu_baseline_config1"] = cpu_baseline(1, 1, "reference")                   # configs[0]: 10 s vs 60 s, one chunk
    emit(json.dumps(out))
    R.close()


def side_measurements(am, device, algo, needle, params, hays, s, h, steps):
    """Numbers that belong next to the headline (never `value`): the worst case of the
    sparse-score path, signals whose scores are not white, and the host-buffer entry."""
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
                                               "one window; a single synchronous call"}
    small.free()
    # (2) signals whose scores are not white (what speech or music against a jingle looks like), each as
    # single calls and as a batch of four (the production shape: the pick of haystack k beside the
    # transforms of k + 1)
    for name, maker in NON_WHITE.items():
        nw_needle, nw_algo, nw_hay, plants, note = maker(am, device, s, h)
        res = nw_algo.match_device(nw_hay.ptr, h, params)
        ok = [p.start for p in res] == plants
        with am.Profile(device) as prof:
            t = timed(lambda: nw_algo.match_device(nw_hay.ptr, h, params), max(2, n // 2))
            tk = {kn: prof.query(kn)[0] / max(prof.query(kn)[1], 1) for kn in KN}
        tb = timed(lambda: nw_algo.match_batch_device([nw_hay.ptr] * 4, [h] * 4, params, cap_per_hay=16), max(2, n // 2)) / 4
        out[name] = {"value": h / tb, "unit": "samples/s", "ms_per_haystack": tb * 1e3, "ms_per_haystack_single_calls": t * 1e3,
                     "offsets_ok": ok, "n_peaks": len(res), "kernel_ms_single_call": tk, "note": note}
        nw_hay.free()
        nw_algo.close()
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
    # (3b) the same haystack pushed in 32 MB pieces (am_match_stream_*): the transforms of the block pairs that
    # have arrived run beside the copy of the next piece
    st = am.MatchStream(algo, params, h)
    piece = 8 << 20
    for rep in range(2):
        t0 = time.perf_counter()
        for off in range(0, h, piece):
            st.push(host[off:off + piece])
        pk = st.finish()
        te = time.perf_counter() - t0
    assert [p.start for p in pk] == plant_offsets(k0)
    st.close()
    out["end_to_end_stream_push"] = {"value": h / te, "unit": "samples/s",
                                     "note": "am_match_stream_begin / push (8 M samples per push from pageable host memory) / finish: "
                                             "every block pair is transformed as soon as its samples have arrived"}
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
    # (5) the host feed in numbers (host/pushbench.cpp, the C ABI from a C++ program): pageable against pinned buffers,
    # pushes of 8 M, 64 K and the decoder's 1152 samples, CPU time of the pushing thread per GB
    exe = os.path.join(ROOT, "audio-matcher_amd", "bin", "pushbench")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
            out["host_feed"] = json.loads(r.stdout) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
        except (subprocess.TimeoutExpired, ValueError) as e:
            out["host_feed"] = {"error": str(e)[:300]}
    return out


def _upload(am, device, needle, hay, s):
    nbuf = am.DeviceBuffer.from_numpy(device, needle)
    algo = am.HipConvolve.from_device(device, nbuf.ptr, s)
    hbuf = am.DeviceBuffer.from_numpy(device, hay)
    return nbuf, algo, hbuf


def make_tonal(am, device, s, h):
    """needle = noise + DC offset + 440 Hz tone; haystack = noise + the same tone + a slow drift
    (240 s period) + 6 planted needles.  The score array then follows the drift (amplitude
    ~0.13, monotone inside every 60 s chunk) with a 440 Hz ripple on top (amplitude ~0.03):
    thousands of ripple maxima per chunk pass the necessary height test, none qualifies.
    Built on the host in f32 (one haystack) and uploaded."""
    import numpy as np
    rng = np.random.default_rng(5)
    t = np.arange(h, dtype=np.float64)
    tone = (0.0437 * np.sin(2 * np.pi * 440.0 / SR * t)).astype(np.float32)
    drift = (0.04 * np.sin(2 * np.pi * t / (240.0 * SR))).astype(np.float32)
    del t
    needle = rng.uniform(-0.25, 0.25, s).astype(np.float32) + tone[:s] + np.float32(0.1)
    hay = rng.uniform(-0.25, 0.25, h).astype(np.float32) + tone + drift
    del tone, drift
    plants = [t for t in plant_offsets(0) if t + s <= h]
    for p0 in plants:
        hay[p0:p0 + s] += needle
    return (*_upload(am, device, needle, hay, s), plants,
            "needle with a DC offset and a 440 Hz tone, haystack with the tone and a 240 s drift: the score array drifts by "
            "+-0.13 with a +-0.03 ripple; exact results")


def _ar1(rng, n, rho, amp):
    """AR(1) noise x[i] = rho x[i-1] + e[i] (a one-pole low-pass: the spectrum of speech / music is
    far from white), scaled to a peak of about `amp`."""
    import numpy as np
    from scipy.signal import lfilter
    e = rng.standard_normal(n).astype(np.float32)
    x = lfilter([1.0], [1.0, -rho], e).astype(np.float32)
    x *= np.float32(amp / 4.0 / np.sqrt(1.0 / (1.0 - rho * rho)))
    return x


def make_ar1(am, device, s, h):
    """needle and haystack are both AR(1) noise (rho = 0.95: a -3 dB corner near 360 Hz at 44.1 kHz);
    the score array is then strongly correlated from lag to lag (broad bumps instead of white noise)."""
    import numpy as np
    rng = np.random.default_rng(7)
    needle = _ar1(rng, s, 0.95, 0.5)
    hay = _ar1(rng, h, 0.95, 0.5)
    plants = [t for t in plant_offsets(0) if t + s <= h]
    for p0 in plants:
        hay[p0:p0 + s] += needle
    return (*_upload(am, device, needle, hay, s), plants,
            "AR(1) noise (rho 0.95) for needle and haystack: a coloured, lag-correlated score array; exact results")


def make_speechlike(am, device, s, h):
    """AR(1) noise under a slow amplitude envelope (syllable-rate 4 Hz modulation times a 20 s loud /
    quiet pattern): the background level of the scores varies by an order of magnitude along the
    haystack, as with speech or music."""
    import numpy as np
    rng = np.random.default_rng(11)
    t = np.arange(h, dtype=np.float32) / np.float32(SR)
    env = (0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t)) * (0.15 + 0.85 * (np.sin(2 * np.pi * t / 20.0) > 0))
    del t
    needle = _ar1(rng, s, 0.9, 0.5)
    hay = _ar1(rng, h, 0.9, 0.6) * env.astype(np.float32)
    del env
    plants = [t for t in plant_offsets(0) if t + s <= h]
    for p0 in plants:
        hay[p0:p0 + s] += needle
    return (*_upload(am, device, needle, hay, s), plants,
            "AR(1) noise (rho 0.9) under a 4 Hz x 20 s loud / quiet envelope (speech-like level changes); exact results")


def make_tonal_hits(am, device, s, h):
    """A tonal needle (a 440 Hz tone under a slow envelope, a little noise) planted into white noise: the
    background scores are tiny, but every hit comes with the needle's autocorrelation -- lobes of -0.98 / +0.96
    half a period (50 samples) either side of the peak.  A chunk with a hit has a minimum near -1 that all
    but one of the K3 tiles never sample: its write-threshold certificate fails."""
    import numpy as np
    rng = np.random.default_rng(13)
    t = np.arange(s, dtype=np.float64)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * t / s)
    needle = (0.2 * env * np.sin(2 * np.pi * 440.0 / SR * t)).astype(np.float32) + rng.uniform(-0.01, 0.01, s).astype(np.float32)
    hay = rng.uniform(-0.25, 0.25, h).astype(np.float32)
    plants = [t0 for t0 in plant_offsets(0) if t0 + s <= h]
    for p0 in plants:
        hay[p0:p0 + s] += needle
    return (*_upload(am, device, needle, hay, s), plants,
            "a tonal needle in white noise: every hit brings autocorrelation lobes of -0.98 with it (chunk minimum near -1 in the "
            "chunks with a hit, background +-0.01): those chunks fail the write-threshold certificate and are redone on the device; "
            "exact results")


NON_WHITE = {"non_white_signal": make_tonal, "non_white_ar1": make_ar1, "non_white_speechlike": make_speechlike,
             "hits_with_negative_lobes": make_tonal_hits}


if __name__ == "__main__":
    main()
