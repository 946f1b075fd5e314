"""Multi-GPU execution of the hot path: one process per GPU (`torch.distributed`, backend "nccl" =
RCCL on ROCm, "gloo" in the CPU tests).

The path shards at depth-map granularity only: `eval.py:213` iterates reference views
with no cross-iteration state, so every rank runs the full single-GPU engine on its own subset of
reference views with replicated weights (0.93 M parameters) and NO collective on the data path.
Collectives appear only at the edges: a barrier + MAX all-reduce of the wall time (bench.py) and an
optional gather of the per-view results to rank 0.

The view-sharded variant named by BASELINE configs 4/5 (each rank warps a subset of the source
views, one all_reduce(SUM) of the sum / sum-of-squares accumulators per level) is implemented in
`view_sharded_variance`: the accumulators are linear in the views, so the exchange is exact up to
fp32 summation order.  It moves 2*C*D*h*w*4 bytes per level through a ring that is bound by one
xGMI link (~153 GB/s), i.e. ~10x-100x the time of building the same level locally from HBM - it is
provided for completeness and measured honestly, not used by default.
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


def view_shard(n_src_views, rank, world):
    """Source views (1-based view ids 1..V-1) rank `rank` warps in the view-sharded variant."""
    return [v for v in range(1, n_src_views + 1) if (v - 1) % world == rank]


def view_sharded_variance(partial_sums_fn, feats, proj_mats, depth_values, group=None):
    """Variance cost volume with the source views split over the ranks of `group`.

    partial_sums_fn(feats, proj_mats, depth_values, views, include_ref) -> (sum, sq) computes
    sum_v warped_v and sum_v warped_v**2 over `views` (plus ref, ref**2 when include_ref) with the
    local engine.  The two accumulators are all-reduced (SUM) and every rank finalises
    var = sq/V - (sum/V)**2 (mvsnet.py:167)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    V = feats.shape[1]
    s, q = partial_sums_fn(feats, proj_mats, depth_values, view_shard(V - 1, rank, world), rank == 0)
    if world > 1:
        buf = torch.stack([s, q])  # one collective per level instead of two
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        s, q = buf[0], buf[1]
    return q.div(V).sub(s.div(V).pow(2))
