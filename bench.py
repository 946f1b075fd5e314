#!/usr/bin/env python
"""Benchmark of the hot path: depth-maps/sec of CascadeMVSNet.forward on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass (FeatureNet + 3 cascade levels) over one batch of --batch reference views (default 2, the
reference's DTU training batch; `--batch 1` is its eval.py loop and is ALSO measured and printed as "batch1") with
their source views: DTU 640x512, 3 views, n_depths [8,32,48], variance cost volume, fp32, synthetic inputs already
resident in HBM, random-init weights.  The timed steps replay the forward as one hipGraph (casmvsnet_pl_amd/graph.py;
`--no-graph` launches kernel by kernel), and --streams (default 2) independent forwards are in flight per GPU, each on
its own HIP stream: a step is then one round of all of them (reference views are independent, eval.py:213); the
single-stream figure is measured and printed beside it ("single_stream").

--mode replica (default): with N GPUs every rank processes its own depth maps (the path shards at depth-map
granularity, SURVEY 8e: no data-path collective) -> weak scaling; value = depth maps all ranks produced / max-over-ranks
wall time.  --mode view_sharded (BASELINE configs 4/5): ALL ranks work on the same depth maps, each warps its share of
the source views and the sum / sum-of-squares volumes are all-reduced over RCCL once per level -> strong scaling;
value = depth maps / max-over-ranks wall time.

Prints ONE JSON line (rank 0).  The `roofline*` objects come from HIP events recorded on the launch stream around
every kernel in an instrumented eager pass over the same inputs right after the timed steps (events cannot be recorded
into a graph replay; same kernels, same shapes); `cpu_baseline` is the oracle (a CPU port of the reference's forward,
oracle/cpu_restatement.py) timed on this host's cores at its best thread count.
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from casmvsnet_pl_amd import ABN, CascadeMVSNet  # noqa: E402
from casmvsnet_pl_amd.graph import ConcurrentForwards, GraphedForward  # noqa: E402
from casmvsnet_pl_amd.profiling import StageTimer  # noqa: E402
from casmvsnet_pl_amd.synthetic import CONFIGS, config_inputs, randomize_state_dict  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak (MI355X_MICROARCH.md)
LAYER_NAMES = ["conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11", "prob"]
HEADLINE = "dtu_640x512_v3_var"


def algorithmic_work(H, W, V, G, n_depths, B=1):
    """Per-depth-map algorithmic bytes / FLOPs, per level."""
    work = {}
    for l in range(3):
        C, D = 8 * 2 ** l, n_depths[l]
        h, w = H // 2 ** l, W // 2 ** l
        n = D * h * w
        cin = G if G > 1 else C
        cout_vol = G if G > 1 else C
        work[l] = {
            "costvol_bytes": 4 * B * (V * C * h * w + D * h * w + cout_vol * n),
            "homo_warp_bytes": 4 * B * (C * h * w + D * h * w + C * n) + 48 * B,   # the un-fused op, one source view
            "softmax_bytes": 4 * B * (2 * n + 2 * h * w),
            "conv0_flops": 2 * 27 * cin * 8 * n * B,
            "costreg_flops": (2 * 27 * cin * 8 + 6480) * n * B,
        }
    return work


def feature_flops(H, W):
    """FeatureNet FLOPs per image (mvsnet.py:14-34): 2 * k*k * cin * cout per output pixel."""
    hw = H * W
    full = 2 * hw * (9 * 3 * 8 + 9 * 8 * 8 + 8 * 32 + 9 * 32 * 8)                      # conv0.0/1, lat0, smooth0
    half = 2 * (hw // 4) * (25 * 8 * 16 + 2 * 9 * 16 * 16 + 16 * 32 + 9 * 32 * 16)      # conv1.*, lat1, smooth1
    quarter = 2 * (hw // 16) * (25 * 16 * 32 + 2 * 9 * 32 * 32 + 32 * 32)               # conv2.*, toplayer
    return full + half + quarter


def pmc_traffic(kernel_prefix, batch):
    """HBM-side bytes per launch of one kernel from the newest committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    need a pass each and cannot be collected inside a timed run; tools/gpu_final.sh collects them in the same gpurun
    call as the bench line it commits, at batch 2): mean over the kernel's launches, read bytes corrected x2 as
    MI355X_MICROARCH.md prescribes."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if batch != 2 or not files:
        return None, "PMC passes are collected at --batch 2 on the default config only"
    path = files[-1]
    prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
    rows = [r for r in json.load(open(path)) if r["kernel"].startswith(prefixes)]
    if not rows:
        return None, "kernel not in " + os.path.relpath(path, ROOT)
    n = sum(r["launches"] for r in rows)
    mb = sum((r["read_mb_corrected"] + r["write_mb"]) * r["launches"] for r in rows) / n
    return mb * 1e6, f"bytes per launch (read + write, mean over the 3 cascade levels) from {os.path.relpath(path, ROOT)}"


def cpu_baseline(cfg_name):
    """Oracle (CPU port of the reference forward) on the same synthetic workload, at the best of a sweep over the
    host's thread count (all 256 hardware threads of the GPU box are 25x slower than 8: oversubscription)."""
    from oracle import cpu_restatement as R
    H, W, V, G, n_depths, ratios, _ = CONFIGS[cfg_name]
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    sd = randomize_state_dict(model.state_dict(), seed=0)
    imgs, proj, dmin, dint = config_inputs(cfg_name, 1, seed=0)
    ncpu = os.cpu_count() or 8
    old = torch.get_num_threads()
    sweep = sorted({t for t in (4, 8, 16, 32, 64) if t <= ncpu})   # beyond 64 threads torch's CPU ops only lose (256 threads: 54 s per forward)
    best, per_threads = None, {}
    t_start = time.perf_counter()
    for i, nt in enumerate(sweep):
        torch.set_num_threads(nt)
        if i == 0:
            R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)  # warm-up (allocator, op dispatch)
        t0 = time.perf_counter()
        R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)
        per_threads[nt] = time.perf_counter() - t0
        if best is None or per_threads[nt] < per_threads[best]:
            best = nt
        if time.perf_counter() - t_start > 40.0:
            break
    torch.set_num_threads(best)
    times = [per_threads[best]]
    for _ in range(2):
        t0 = time.perf_counter()
        R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(old)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "depth-maps/s", "cores": best, "kind": "port",
            "sample": f"3 timed forwards of ONE depth map each (median {med:.3f} s) on the same {cfg_name} inputs / weights, torch CPU "
                      f"fp32 at {best} threads = the best of a sweep {({k: round(v, 2) for k, v in per_threads.items()})} s over "
                      f"{ncpu} hardware threads"}


def stock_pytorch_rocm(cfg_name, dev):
    """SURVEY 8(d)'s second comparison row: the SAME restated reference forward, executed on the MI355X by stock
    PyTorch-ROCm operators (MIOpen convolutions, ATen grid_sample / batch_norm / softmax) - what a plain `model.cuda()` of
    the reference gives.  Part of the baseline leg like cpu_baseline (the only place bench.py touches oracle/).  The
    first call spends about a minute in MIOpen's kernel search; it is not timed."""
    from oracle import cpu_restatement as R
    H, W, V, G, n_depths, ratios, _ = CONFIGS[cfg_name]
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    sd = {k: v.to(dev) for k, v in randomize_state_dict(model.state_dict(), seed=0).items()}
    imgs, proj, dmin, dint = config_inputs(cfg_name, 1, seed=0)
    imgs, proj = imgs.to(dev), proj.to(dev)
    old = torch.get_default_device() if hasattr(torch, "get_default_device") else None
    torch.set_default_device(dev)   # the restatement creates its grids / plane indices on the default device
    try:
        with torch.no_grad():
            for _ in range(3):
                R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)
            torch.cuda.synchronize()
            times = []
            for _ in range(7):
                t0 = time.perf_counter()
                R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
    finally:
        torch.set_default_device(old if old is not None else "cpu")
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "depth-maps/s", "ms_per_forward": 1e3 * med, "kind": "port on stock PyTorch-ROCm operators",
            "sample": f"median of 7 forwards of ONE depth map ({cfg_name}) after 3 warm-ups, torch {torch.__version__} on the same GPU"}


def timed_steps(step, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    return time.perf_counter() - t0, out


def homo_warp_roofline(dev, H, W, n_depths, B):
    """The un-fused op models/modules.py:52-92 (north_star names its HBM-roofline fraction) at the three level
    shapes: HIP events around 10 calls each, algorithmic bytes 4 B (C h w + D h w + C D h w)."""
    from casmvsnet_pl_amd import ops
    from casmvsnet_pl_amd.synthetic import make_inputs
    _, proj, dmin, dint = make_inputs(B, 2, H, W, seed=0)
    work = algorithmic_work(H, W, 2, 1, n_depths, B)
    per_level, tot_b, tot_ms = {}, 0.0, 0.0
    for l in range(3):
        C, D, h, w = 8 * 2 ** l, n_depths[l], H >> l, W >> l
        src = torch.randn(B, h, w, C, device=dev)   # pixel-major, as FeatureNet hands it to the engine
        P = proj[:, 0, l].contiguous().to(dev)
        step = dint * 2 ** l
        k = torch.arange(D, device=dev, dtype=torch.float32).view(1, D, 1, 1)
        depth = (680.0 - D / 2 * step + 40.0 * torch.sin(torch.linspace(0, 6.0, w, device=dev)).view(1, 1, 1, w) + k * step).expand(B, D, h, w).contiguous()
        fn = lambda: ops.homo_warp_nhwc(src, P, depth)
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        per_level[str(l)] = work[l]["homo_warp_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        tot_b += work[l]["homo_warp_bytes"]
        tot_ms += ms
    ach = tot_b / (tot_ms * 1e-3) / 1e9
    return {"kernel": "homo_warp (un-fused op, casmvs_homo_warp_nhwc_f32; one launch per level shape)", "bound": "hbm",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "per_level_frac": per_level, "avg_launch_ms": tot_ms / 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=HEADLINE, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=2,
                    help="depth maps per step per GPU (reference views batched like the reference's train.py --batch_size 2; "
                         "--batch 1 = the reference's eval.py loop, always measured too)")
    ap.add_argument("--mode", default="replica", choices=["replica", "view_sharded"])
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying one hipGraph")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent forwards in flight per GPU, one HIP stream + hipGraph each (a step = one round of all of "
                         "them; 1 = a single forward per step).  The single-stream figures are printed as well.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stock-pytorch", action="store_true",
                    help="also time the restated reference forward on stock PyTorch-ROCm operators on this GPU (adds ~1 min of MIOpen search)")
    ap.add_argument("--no-events", action="store_true", help="skip the instrumented pass (no roofline objects)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="the instrumented pass records its ~90 HIP events on every n-th of its K kernel-by-kernel steps (an event "
                         "costs ~3 us of GPU time; on every step they would inflate the step by ~10 %% and starve the launch queue)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the extra batch-1 measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.mode == "view_sharded":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    H, W, V, G, n_depths, ratios, _ = CONFIGS[args.config]
    view_sharded = args.mode == "view_sharded"

    def build(B):
        model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
        randomize_state_dict(model.state_dict(), seed=0)
        model = model.to(dev).eval()
        # replica: every rank works on its own depth maps (different seeds -> different images / cameras);
        # view_sharded: all ranks share the depth maps and split their source views
        imgs, proj, dmin, dint = config_inputs(args.config, B, seed=0 if view_sharded else rank)
        if view_sharded:
            model.view_shard_group = dist.group.WORLD
        return model, imgs.to(dev), proj.to(dev), dmin, dint

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(B, steps, warmup, streams=1):
        model, imgs, proj, dmin, dint = build(B)
        for _ in range(warmup):
            model(imgs, proj, dmin, dint)
        use_graph = not args.no_graph and not view_sharded   # a collective inside a capture is not attempted
        if use_graph and streams > 1:
            cf = ConcurrentForwards(model, imgs, proj, dmin, dint, n_streams=streams)
            step = lambda: cf.run()[-1]
        elif use_graph:
            gf = GraphedForward(model, imgs, proj, dmin, dint)
            step = lambda: gf(imgs, proj)
        else:
            step = lambda: model(imgs, proj, dmin, dint)
        elapsed, out = timed_steps(step, steps, barrier)
        if dist is not None:
            te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())
        assert torch.isfinite(out["depth_0"]).all()
        return model, (imgs, proj, dmin, dint), elapsed, use_graph

    B = args.batch
    NS = 1 if (view_sharded or args.no_graph) else max(1, args.streams)
    model, inputs, elapsed, used_graph = measure(B, args.steps, args.warmup, NS)
    K = args.steps
    maps = (1 if view_sharded else world) * B * NS * K
    line = None
    if rank == 0:
        line = {
            "metric": "depth-maps/sec at 640x512, 3 views, n_depths=[8,32,48]" if args.config == HEADLINE
                      else f"depth-maps/sec ({args.config})",
            "value": maps / elapsed, "unit": "depth-maps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong" if view_sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "H": H, "W": W, "views": V, "n_depths": list(n_depths),
                       "interval_ratios": list(ratios), "num_groups": G, "depth_maps_per_step_per_gpu": B * NS,
                       "batch_per_forward": B, "concurrent_forwards_per_gpu": NS,
                       "depth_interval": inputs[3], "init_depth_min": inputs[2],
                       "launch": (f"{NS} independent forwards per step, each one hipGraph replay on its own HIP stream" if NS > 1 else
                                  "one hipGraph replay per step") if used_graph else "kernel by kernel",
                       "parallelism": (f"view-sharded x{world}: source views split over the ranks, one RCCL all-reduce of the sum / "
                                       "sum-of-squares volumes per level, every rank regularises") if view_sharded else
                                      f"replica x{world} (one depth map stream per GPU, no data-path collective)",
                       "feature_net": "HIP MFMA kernels (casmvs_featurenet_forward_f32)"},
        }

    # ---- instrumented eager pass: HIP events around every kernel (same model, same inputs) ------------------------
    if not args.no_events:
        every = max(1, args.event_every)
        n_ev = (K + every - 1) // every
        timer = StageTimer()
        timer.reserve((2 * 13 + 14 + 36) * n_ev)
        barrier()
        t0 = time.perf_counter()
        for i in range(K):   # K kernel-by-kernel steps, events on every `every`-th one: the GPU stays busy in between
            model.set_timer(timer if i % every == 0 else None)
            model(*inputs)
        barrier()
        eager_ms = 1e3 * (time.perf_counter() - t0) / K
        model.set_timer(None)
        if rank == 0:
            summ = timer.summary(LAYER_NAMES)
            work = algorithmic_work(H, W, V, G, n_depths, B)
            per_step = {k: v["ms"] / n_ev for k, v in summ.items()}
            # dominant kernel: conv16db_kernel<PX> = CostRegNet.conv0 (3 launches per step)
            conv0_ms = sum(summ[f"costreg_{l}/conv0"]["ms"] for l in range(3))
            conv0_flops = sum(work[l]["conv0_flops"] for l in range(3)) * n_ev
            ach = conv0_flops / (conv0_ms * 1e-3) / 1e12
            traffic, traffic_note = pmc_traffic("conv16db_kernel<2, 4, 4, 4, 4, 32", B if args.config == HEADLINE else None)
            line["roofline"] = {"kernel": "conv16db_kernel<PX> (CostRegNet.conv0: Cout 8, stride 1; 3 launches per step)",
                                "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                                "traffic_note": traffic_note + "; algorithmic bytes of the same launches: 384e6",
                                "avg_launch_ms": conv0_ms / (3 * n_ev)}
            cr_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("costreg_"))
            cr_flops = sum(work[l]["costreg_flops"] for l in range(3)) * n_ev
            line["roofline_costreg"] = {"kernel": "all 33 CostRegNet launches", "bound": "mfma",
                                        "achieved": cr_flops / (cr_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": cr_flops / (cr_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                        "ms_per_depth_map": cr_ms / n_ev / B}
            cv_ms = sum(summ[f"costvol_{l}"]["ms"] for l in range(3))
            cv_bytes = sum(work[l]["costvol_bytes"] for l in range(3)) * n_ev
            cv_traffic, cv_note = pmc_traffic(("costvol_lds_kernel", "costvol_nhwc_kernel"), B if args.config == HEADLINE else None)
            line["roofline_costvol"] = {"kernel": "fused homo_warp + aggregation (3 launches: costvol_lds_kernel at C = 8 / 16, "
                                                  "costvol_nhwc_kernel at C = 32 and for group-wise correlation)",
                                        "bound": "hbm", "achieved": cv_bytes / (cv_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": cv_bytes / (cv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "traffic": cv_traffic, "traffic_note": cv_note + f"; algorithmic: {cv_bytes / n_ev / 3:.4g}",
                                        "ms_per_depth_map": cv_ms / n_ev / B,
                                        "per_level_frac": {str(l): work[l]["costvol_bytes"] * n_ev / (summ[f"costvol_{l}"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS for l in range(3)}}
            sm_ms = sum(summ[f"softmax_{l}"]["ms"] for l in range(3))
            sm_bytes = sum(work[l]["softmax_bytes"] for l in range(3)) * n_ev
            line["roofline_softmax"] = {"kernel": "softmax_regress_kernel (3 launches)", "bound": "hbm",
                                        "achieved": sm_bytes / (sm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": sm_bytes / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            ft_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("feature/"))
            if ft_ms > 0:
                ft_flops = feature_flops(H, W) * V * B * n_ev
                line["roofline_feature"] = {"kernel": "all 13 FeatureNet launches", "bound": "mfma",
                                            "achieved": ft_flops / (ft_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                            "unit": "TFLOP/s", "frac": ft_flops / (ft_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                            "ms_per_depth_map": ft_ms / n_ev / B}
            line["roofline_homo_warp"] = homo_warp_roofline(dev, H, W, n_depths, B)
            line["stage_ms_per_step"] = {k: round(v, 4) for k, v in per_step.items()}
            line["instrumented_pass"] = {"steps": K, "event_sampled_steps": n_ev, "ms_per_step": eager_ms,
                                         "note": f"kernel-by-kernel launches right after the timed steps, ~90 HIP events on every {every}-th step"}
    del model
    # ---- the reference's eval.py loop: one reference view per step ---------------------------------------------------
    if NS > 1:
        _, _, els, gs = measure(B, K, max(2, args.warmup // 2), 1)
        if rank == 0:
            line["single_stream"] = {"value": world * B * K / els, "unit": "depth-maps/s", "ms_per_step": 1e3 * els / K, "steps": K,
                                     "note": f"one forward of batch {B} per step (one stream, one hipGraph replay)"}
    if B != 1 and not args.no_batch1:
        _, _, el1, g1 = measure(1, K, max(2, args.warmup // 2), 1)
        if rank == 0:
            line["batch1"] = {"value": (1 if view_sharded else world) * K / el1, "unit": "depth-maps/s", "ms_per_step": 1e3 * el1 / K,
                              "steps": K, "launch": "one hipGraph replay per step" if g1 else "kernel by kernel",
                              "note": "eval.py:213-222 processes one reference view per forward (one stream)"}
        if NS > 1:
            _, _, el1s, _ = measure(1, K, max(2, args.warmup // 2), NS)
            if rank == 0:
                line["batch1"]["concurrent"] = {"value": world * NS * K / el1s, "unit": "depth-maps/s", "ms_per_step": 1e3 * el1s / K,
                                                "note": f"{NS} single-view forwards in flight, one stream each"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config)
        if world == 1 and args.stock_pytorch:
            line["stock_pytorch_rocm"] = stock_pytorch_rocm(args.config, dev)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
