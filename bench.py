#!/usr/bin/env python
"""Benchmark of the hot path: depth-maps/sec of CascadeMVSNet.forward on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass (FeatureNet + 3 cascade levels) over one batch of --batch reference views
(default 2) with their source views: DTU 640x512, 3 views, n_depths [8,32,48], variance cost volume,
fp32, synthetic inputs already resident in HBM, random-init weights.  With N GPUs every rank processes its own
depth maps (the path shards at depth-map granularity, SURVEY 8e: no data-path collective) ->
weak scaling; value = depth maps all ranks produced / max-over-ranks wall time.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events recorded on the
launch stream inside the timed region; `cpu_baseline` is the oracle (a CPU port of the
reference's forward, oracle/cpu_restatement.py) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from casmvsnet_pl_amd import ABN, CascadeMVSNet  # noqa: E402
from casmvsnet_pl_amd.profiling import StageTimer  # noqa: E402
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA dense peak (MI355X_MICROARCH.md)

CONFIGS = {
    # name: (H, W, V, num_groups, n_depths, interval_ratios)
    "dtu_640x512_v3_var": (512, 640, 3, 1, (8, 32, 48), (1.0, 2.0, 4.0)),
    "dtu_640x512_v3_gwc8": (512, 640, 3, 8, (8, 32, 48), (1.0, 2.0, 4.0)),
    "dtu_1152x864_v5_var": (864, 1152, 5, 1, (8, 32, 48), (1.0, 2.0, 4.0)),
    "blended_768x576_v7_var": (576, 768, 7, 1, (8, 32, 48), (1.0, 2.0, 4.0)),
}
LAYER_NAMES = ["conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11", "prob"]


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
    """HBM-side bytes per launch of one kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in their own runs, tools/gpu_final.sh -> profiles/r01_final_pmc_traffic.json,
    at batch 2): average over the kernel's launches, read bytes corrected x2 as MI355X_MICROARCH.md prescribes."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")
    if batch != 2 or not os.path.isfile(path):
        return None, "PMC passes were collected at --batch 2 on the default config only"
    rows = [r for r in json.load(open(path)) if r["kernel"].startswith(kernel_prefix)]
    if not rows:
        return None, "kernel not in " + os.path.relpath(path, ROOT)
    n = sum(r["launches"] for r in rows)
    mb = sum((r["read_mb_corrected"] + r["write_mb"]) * r["launches"] for r in rows) / n
    return mb * 1e6, ("bytes per launch (read + write, mean over the 3 cascade levels) from profiles/r01_final_pmc_traffic.json; "
                      "algorithmic bytes of the same launches: 384e6")


def cpu_baseline(cfg_name, repeats=3):
    """Oracle (CPU port of the reference forward) on the same synthetic workload, host cores."""
    from oracle import cpu_restatement as R
    H, W, V, G, n_depths, ratios = CONFIGS[cfg_name]
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    sd = randomize_state_dict(model.state_dict(), seed=0)
    imgs, proj, dmin, dint = make_inputs(1, V, H, W, seed=0)
    R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)  # warm-up
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "depth-maps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{repeats} timed forwards of ONE depth map each (median {med:.3f} s) on the same {cfg_name} inputs / weights "
                      f"after 1 warm-up, torch CPU fp32, {torch.get_num_threads()} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="dtu_640x512_v3_var", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=2,
                    help="depth maps per step per GPU (reference views batched like the reference's train.py --batch_size 2; "
                         "--batch 1 = the reference's eval.py loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--event-every", type=int, default=4,
                    help="record the per-kernel HIP events on every n-th timed step (an event costs ~3 us of GPU time; "
                         "~90 per step would inflate the step by ~10 %%)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    H, W, V, G, n_depths, ratios = CONFIGS[args.config]
    B = args.batch
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=0)
    model = model.to(dev).eval()
    # every rank works on its own depth maps (different seeds -> different images / cameras)
    imgs, proj, dmin, dint = make_inputs(B, V, H, W, seed=rank)
    imgs, proj = imgs.to(dev), proj.to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model(imgs, proj, dmin, dint)
    timer = None if args.no_events else StageTimer()
    n_ev = 0
    if timer is not None:  # 2 events per stage range (1 + 4 x 3 stages) + 14 FeatureNet + 3 x 12 CostRegNet per sampled step
        timer.reserve((2 * 13 + 14 + 36) * ((args.steps + max(args.event_every, 1) - 1) // max(args.event_every, 1)))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = timer is not None and i % max(args.event_every, 1) == 0
        model.set_timer(timer if sampled else None)
        n_ev += int(sampled)
        out = model(imgs, proj, dmin, dint)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    assert torch.isfinite(out["depth_0"]).all()

    if rank == 0:
        K = args.steps
        value = world * B * K / elapsed
        line = {
            "metric": "depth-maps/sec at 640x512, 3 views, n_depths=[8,32,48]" if args.config == "dtu_640x512_v3_var"
                      else f"depth-maps/sec ({args.config})",
            "value": value, "unit": "depth-maps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "H": H, "W": W, "views": V, "n_depths": list(n_depths),
                       "interval_ratios": list(ratios), "num_groups": G, "depth_maps_per_step_per_gpu": B,
                       "parallelism": f"replica x{world} (one depth map stream per GPU, no data-path collective)",
                       "feature_net": "HIP MFMA kernels (casmvs_featurenet_forward_f32)"},
        }
        if timer is not None:
            summ = timer.summary(LAYER_NAMES)
            work = algorithmic_work(H, W, V, G, n_depths, B)
            K_all, K = K, n_ev   # the event-derived figures below are averages over the n_ev sampled steps
            per_step = {k: v["ms"] / K for k, v in summ.items()}
            # dominant kernel: conv3d_kernel<S1, Cout 8> = CostRegNet.conv0 (3 launches per depth map)
            conv0_ms = sum(summ[f"costreg_{l}/conv0"]["ms"] for l in range(3))
            conv0_flops = sum(work[l]["conv0_flops"] for l in range(3)) * K
            ach = conv0_flops / (conv0_ms * 1e-3) / 1e12
            traffic, traffic_note = pmc_traffic("conv16db_kernel<2, 4, 4, 4, 4, 32", B if args.config == "dtu_640x512_v3_var" else None)
            line["roofline"] = {"kernel": "conv16db_kernel<PX> (CostRegNet.conv0: Cout 8, stride 1; 3 launches per step)",
                                "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                                "avg_launch_ms": conv0_ms / (3 * K)}
            cr_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("costreg_"))
            cr_flops = sum(work[l]["costreg_flops"] for l in range(3)) * K
            line["roofline_costreg"] = {"kernel": "all 33 CostRegNet launches", "bound": "mfma",
                                        "achieved": cr_flops / (cr_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": cr_flops / (cr_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                        "ms_per_depth_map": cr_ms / K / B}
            cv_ms = sum(summ[f"costvol_{l}"]["ms"] for l in range(3))
            cv_bytes = sum(work[l]["costvol_bytes"] for l in range(3)) * K
            line["roofline_costvol"] = {"kernel": "costvol_kernel (fused homo_warp + aggregation, 3 launches)",
                                        "bound": "hbm", "achieved": cv_bytes / (cv_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": cv_bytes / (cv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "traffic": None, "ms_per_depth_map": cv_ms / K / B,
                                        "per_level_frac": {str(l): work[l]["costvol_bytes"] * K / (summ[f"costvol_{l}"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS for l in range(3)}}
            sm_ms = sum(summ[f"softmax_{l}"]["ms"] for l in range(3))
            sm_bytes = sum(work[l]["softmax_bytes"] for l in range(3)) * K
            line["roofline_softmax"] = {"kernel": "softmax_regress_kernel (3 launches)", "bound": "hbm",
                                        "achieved": sm_bytes / (sm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": sm_bytes / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            ft_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("feature/"))
            if ft_ms > 0:
                ft_flops = feature_flops(H, W) * V * B * K
                line["roofline_feature"] = {"kernel": "all 13 FeatureNet launches", "bound": "mfma",
                                            "achieved": ft_flops / (ft_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                            "unit": "TFLOP/s", "frac": ft_flops / (ft_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                            "ms_per_depth_map": ft_ms / K / B}
            line["stage_ms_per_step"] = {k: round(v, 4) for k, v in per_step.items()}
            line["event_sampled_steps"] = n_ev
            K = K_all
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
