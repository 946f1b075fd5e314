"""Multi-GPU execution of the hot path: one process per GPU (`torch.distributed`, backend "nccl" =
RCCL on ROCm, "gloo" in the CPU tests).

The path shards at depth-map granularity only: `eval.py:213` iterates reference views
with no cross-iteration state, so every rank runs the full single-GPU engine on its own subset of
reference views with replicated weights (0.93 M parameters) and NO collective on the data path.
Collectives appear only at the edges: a barrier + MAX all-reduce of the wall time (bench.py) and an
optional gather of the per-view results to rank 0.

The view-sharded variant named by BASELINE configs 4/5 (each rank warps a subset of the source
views, one all_reduce(SUM) of the sum / sum-of-squares accumulators per level) is implemented in
`view_sharded_cost_volume` on the HIP partial-sum kernels (`CascadeMVSNet.view_shard_group`,
`bench.py --mode view_sharded`): the accumulators are linear in the views, so the exchange is exact up to
fp32 summation order.  It moves 2*C*D*h*w*4 bytes per level through a ring that is bound by one
xGMI link (~153 GB/s), i.e. ~10x-100x the time of building the same level locally from HBM - it is
provided because the configs name it and measured honestly, not used by default.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's RANK / WORLD_SIZE / MASTER_* variables.
    Returns (rank, world_size, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def shard_indices(n_items, rank, world):
    """Indices of the work items (reference views) rank `rank` owns: round-robin, so that every rank
    gets ceil or floor of n/world items and neighbouring views (similar cost) spread over ranks."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


def run_sharded(n_items, process_item, rank=None, world=None, gather=True):
    """Run `process_item(i) -> dict[str, Tensor]` for this rank's items.  With gather=True rank 0
    returns the full list (index order) and the other ranks return None; tensors travel as CPU
    objects (results are small (h, w) maps; this is control-plane traffic, not the data path)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    mine = {i: {k: v.detach().cpu() for k, v in process_item(i).items()} for i in shard_indices(n_items, rank, world)}
    if not gather or world == 1:
        return [mine[i] for i in sorted(mine)] if world == 1 else mine
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    if sorted(merged) != list(range(n_items)):
        raise RuntimeError("run_sharded: gathered item set is not a partition of the work list")
    return [merged[i] for i in range(n_items)]


def max_over_ranks(seconds, device):
    """MAX all-reduce of a wall-clock interval (the bench's timing rule)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def view_range(n_src_views, rank, world):
    """Source views [begin, end) (1-based view ids 1..V-1; view 0 is the reference) that rank `rank` warps in the
    view-sharded build: contiguous, balanced (sizes differ by at most one); ranks beyond the number of source views
    get an empty range."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_src_views, world)
    begin = 1 + rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class HipSweepEngine:
    """The per-rank engine of the view-sharded build: the HIP partial-sum / finalise kernels (include/casmvs.h:
    casmvs_costvol_partial_{var,gwc}_f32, casmvs_costvol_{var,gwc}_finalize_f32)."""

    @staticmethod
    def partial(feats_nhwc, proj_mats, depth_values, begin, end, num_groups, include_ref):
        from . import ops
        return ops.costvol_partial(feats_nhwc, proj_mats, depth_values, begin, end, num_groups, include_ref)

    @staticmethod
    def zeros_like_partial(feats_nhwc, depth_values, num_groups):
        B, V, h, w, C = feats_nhwc.shape
        D = depth_values.shape[1]
        shape = (2, B, C, D, h, w) if num_groups == 1 else (B, num_groups, D, h, w)
        return torch.zeros(shape, dtype=torch.float32, device=feats_nhwc.device)

    @staticmethod
    def finalize(partial, V, num_groups):
        from . import ops
        return ops.costvol_finalize(partial, V, num_groups)


def view_sharded_cost_volume(feats_nhwc, proj_mats, depth_values, num_groups=1, group=None, engine=HipSweepEngine):
    """Cost volume of ONE set of reference views with the source views split over the ranks of `group`
    (models/mvsnet.py:147-167 is the loop being sharded; BASELINE configs 4/5).

    The sums are linear in the source views: rank r warps its views [begin, end) against the full reference
    frustum - rank 0 also adds the reference terms ref / ref^2 - the partial buffers ((2,B,C,D,h,w) = sum and
    sum of squares for the variance; (B,G,D,h,w) for the correlation) are combined with ONE all_reduce(SUM) per level,
    and every rank finalises (var = sq/V - (sum/V)^2, or / (V-1)).  Exact up to fp32 summation order; with one rank it
    equals the fused kernel bit for bit.  feats_nhwc (B,V,h,w,C) pixel-major, all views present on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    V = feats_nhwc.shape[1]
    begin, end = view_range(V - 1, rank, world)
    if begin < end:
        part = engine.partial(feats_nhwc, proj_mats, depth_values, begin, end, num_groups, rank == 0)
    else:   # more ranks than source views: this rank contributes nothing
        part = engine.zeros_like_partial(feats_nhwc, depth_values, num_groups)
    if world > 1:
        dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
    return engine.finalize(part, V, num_groups)
