"""Multi-GPU sharding of a haystack batch: independent haystacks, one process per
GPU, no data-path collective (the reference loops over files serially,
matcher/mod.rs:42; chunks of one file fan out over rayon, audio_matcher.rs:114).
torch.distributed is used only for control: a barrier, the max-over-ranks of a
timed region, and the host-side gather of the (tiny) peak lists."""
from __future__ import annotations


def shard_indices(n_items: int, rank: int, world: int):
    """Haystack k -> rank k mod world: the rule of the C ABI's am_shard_plan,
    which the in-process pool (am_pool_*) uses too (a pure function, no device needed)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    from . import shard_plan
    first, stride, count = shard_plan(n_items, world, rank)
    return [first + i * stride for i in range(count)]


def owner_of(k: int, world: int) -> int:
    return k % world


def max_over_ranks(value: float, dist=None) -> float:
    """MAX all-reduce of a host scalar (the timed region's duration)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local: dict, dist=None) -> dict:
    """Host-side concatenation of per-haystack peak lists {k: peaks}."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        for k, v in p.items():
            if k in merged:
                raise RuntimeError(f"haystack {k} processed by two ranks")
            merged[k] = v
    return merged
