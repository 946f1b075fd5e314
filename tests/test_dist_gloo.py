"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: depth-map sharding + gather, the MAX
timing rule, and the view-sharded variance exchange (linearity of the accumulators), with the
oracle standing in for the per-rank engine (tests only)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, fn_name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, globals()[fn_name](rank, world)))
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:   # join first (with a timeout): a worker that died before its put() must not hang q.get()
        p.join(180)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    res = {}
    while not q.empty():
        k, v = q.get()
        res[k] = v
    assert len(res) == world
    return res


def _case_run_sharded(rank, world):
    from casmvsnet_pl_amd.dist import max_over_ranks, run_sharded, shard_indices

    def process(i):
        return {"depth_0": torch.full((2, 3), float(i)), "owner": torch.tensor(rank)}
    out = run_sharded(7, process, gather=True)
    t = max_over_ranks(1.0 + rank, torch.device("cpu"))
    return {"mine": shard_indices(7, rank, world), "t": t,
            "gathered": None if out is None else [(float(o["depth_0"][0, 0]), int(o["owner"])) for o in out]}


class _OracleEngine:
    """The oracle standing in for the per-rank HIP engine (CPU test of the exchange logic only)."""

    @staticmethod
    def partial(feats_nhwc, proj, depth, begin, end, G, include_ref):
        from oracle import cpu_restatement as R
        feats = feats_nhwc.permute(0, 1, 4, 2, 3)
        B, V, C, h, w = feats.shape
        D = depth.shape[1]
        ref = feats[:, 0].unsqueeze(2).expand(-1, -1, D, -1, -1)
        s = ref.clone() if (include_ref and G == 1) else torch.zeros(B, C, D, h, w)
        q = ref ** 2 if (include_ref and G == 1) else torch.zeros(B, C, D, h, w)
        for v in range(begin, end):
            wv = R.homo_warp(feats[:, v].contiguous(), proj[:, v - 1], depth)
            s, q = s + wv, q + wv ** 2
        if G == 1:
            return torch.stack([s, q])
        return (s * ref).reshape(B, G, C // G, D, h, w).mean(2)

    @staticmethod
    def zeros_like_partial(feats_nhwc, depth, G):
        B, V, h, w, C = feats_nhwc.shape
        D = depth.shape[1]
        return torch.zeros((2, B, C, D, h, w) if G == 1 else (B, G, D, h, w))

    @staticmethod
    def finalize(part, V, G):
        if G == 1:
            return part[1].div(V).sub(part[0].div(V).pow(2))
        return part.div(V - 1)


def _case_view_sharded(rank, world):
    from casmvsnet_pl_amd.dist import view_sharded_cost_volume
    from oracle import cpu_restatement as R
    from casmvsnet_pl_amd.synthetic import make_inputs
    g = torch.Generator().manual_seed(0)
    B, V, C, h, w, D = 1, 4, 8, 16, 24, 6
    feats = torch.randn(B, V, C, h, w, generator=g)
    _, proj, dmin, dint = make_inputs(B, V, h, w, seed=1)
    proj = proj[:, :, 0].contiguous()
    depth = dmin + torch.arange(D).view(1, D, 1, 1) * dint * 4 + torch.zeros(B, D, h, w)
    nhwc = feats.permute(0, 1, 3, 4, 2).contiguous()
    errs = []
    for G in (1, 4):
        got = view_sharded_cost_volume(nhwc, proj, depth, G, engine=_OracleEngine)
        want = R.cost_volume(feats, proj, depth, G)
        errs.append(float((got - want).abs().max() / want.abs().max()))
    return max(errs)


def test_shard_indices_partition():
    from casmvsnet_pl_amd.dist import shard_indices, view_range
    for n in (0, 1, 7, 49):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    for n_src, world in ((6, 4), (4, 8), (2, 2), (1, 3)):   # contiguous, balanced, a partition of 1..n_src
        rs = [view_range(n_src, r, world) for r in range(world)]
        assert sum((list(range(b, e)) for b, e in rs), []) == list(range(1, n_src + 1))
        assert max(e - b for b, e in rs) - min(e - b for b, e in rs) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def test_run_sharded_gathers_in_order_world2():
    res = _spawn("_case_run_sharded")
    assert res[0]["mine"] == [0, 2, 4, 6] and res[1]["mine"] == [1, 3, 5]
    assert res[1]["gathered"] is None
    assert res[0]["gathered"] == [(float(i), i % 2) for i in range(7)]
    assert res[0]["t"] == res[1]["t"] == 2.0  # MAX over ranks


def test_view_sharded_variance_equals_local_world2():
    res = _spawn("_case_view_sharded")
    assert res[0] < 1e-5 and res[1] < 1e-5
