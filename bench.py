#!/usr/bin/env python
"""Benchmark of the hot path: depth-maps/sec of CascadeMVSNet.forward on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass (FeatureNet + 3 cascade levels) over one batch of --batch reference views (default 8: independent
reference views batched like the reference's train.py --batch_size; `--batch 1` is its eval.py loop and is ALSO measured and
printed as "batch1", with its own roofline objects) with their source views: DTU 640x512, 3 views, n_depths [8,32,48], variance
cost volume, float32 tensors, synthetic inputs already resident in HBM, random-init weights.  The timed steps replay the forward as
ONE hipGraph on ONE stream (casmvsnet_pl_amd/graph.py; `--no-graph` launches kernel by kernel).  `--streams N` > 1 puts N independent
forwards of --batch views in flight, each on its own HIP stream with the model's own split-f16 layers (graph.ConcurrentForwards); the default
line carries that launch as `two_streams` (2 x half the batch) and `batch1.two_streams` (2 x 1 view: the eval.py loop two views at a time).

`value` = depth maps of all ranks / the max-over-ranks wall time of EXACTLY K steps (barrier + synchronize on both
sides); `median_ms_per_step` (the median of the timed iterations) comes from one HIP event per step recorded
inside the same timed region.

--mode replica (default): with N GPUs every rank processes its own depth maps (the path shards at depth-map
granularity, SURVEY 8e: no data-path collective) -> weak scaling.  --mode view_sharded (BASELINE configs 4/5): ALL ranks
work on the same depth maps, each warps its share of the source views and the sum / sum-of-squares volumes are
all-reduced over RCCL once per level -> strong scaling.  --mode train: the reference's training step (train.py:99-127:
train-mode forward, SL1 loss, backward, SGD) through the HIP training path; prints `train_step_ms` (metric:
samples/s), no roofline objects.  The default line carries the same measurement as `train_step` (20 hipGraph replays of
the batch-1 training step, ~0.3 s; --no-train-step skips it).

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (CostRegNet.conv0, three launches per step) against the roof that
binds it: HBM - `achieved` = its ALGORITHMIC bytes (input + output volume of each launch) / its HIP-event time, `frac` against 8 TB/s;
`traffic` = the kernel's PMC bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) when the committed PMC passes were collected on THIS build
of the library (sha256 checked), else null; `roofline.mfma` holds the matrix-core view of the same launches (the f16 FLOPs they
execute against the dense f16 peak, and `fp32_equivalent`: the layer's float32 FLOPs against the float32 MFMA peak, the figure earlier
rounds led with - it is NOT a roofline fraction for a kernel that multiplies on the f16 cores).  The `roofline*` objects come from HIP events recorded on the launch stream around
every kernel in an instrumented eager pass over the same inputs right after the timed steps (events cannot be recorded
into a graph replay; same kernels, same shapes); `cpu_baseline` is the reference's own forward (/root/reference +
import shims, kind "reference") when that tree exists, else the oracle (a CPU port of it, oracle/cpu_restatement.py,
kind "port" - the GPU box has no /root/reference), timed on this host's cores at its best thread count.
"""
import argparse
import glob
import hashlib
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
MFMA_F16_PEAK_TFLOPS = 2500.0  # f16 / bf16 MFMA dense peak (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)
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
            # the `prob` head fused with the regression: 8 input channels + hypotheses read, cost + 2 maps written
            "prob_regress_bytes": 4 * B * (8 * n + n + n + 2 * h * w),
            "tail_bytes": 4 * B * (2 * n + 8 * n + n + n + 2 * h * w),   # conv11 in (16 ch at n / 8 voxels), skip, hypotheses, cost, depth + confidence
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


def feature_flops_f32_layers(H, W, fused_conv0=True):
    """The FeatureNet layers that stay on the float32 MFMA in the split-f16 layer set: lat1, toplayer - and conv0.0 / conv0.1 unless they run as the fused
    f16 kernel (fnet_conv0_mm.hip, round 6; FeatureNet.fuse_conv0)."""
    hw = H * W
    return (0 if fused_conv0 else 2 * hw * (9 * 3 * 8 + 9 * 8 * 8)) + 2 * (hw // 4) * (16 * 32) + 2 * (hw // 16) * (32 * 32)


COSTREG_F32_FLOPS_PER_VOXEL = 216 + 216 + 432   # conv5, conv7, `prob`: the CostRegNet layers that stay float32 in the split-f16 layer set (of 6480 + conv0's)


def mixed_mfma_roofline(flops, flops_f32_layers, ms, split):
    """Roofline object of a stage that runs some layers on the f16 matrix cores (float32 operands as two float16 slices: 3 partial products x 4/3 K padding
    = 4 executed FLOPs per algorithmic FLOP) and the rest on the float32 MFMA.  `frac` is quoted against the peak of what EXECUTES: the stage's ideal
    time - executed f16 FLOPs at the dense f16 peak plus float32 FLOPs at the float32 MFMA peak - over its measured time.  The float32-peak RATIO of the
    algorithmic FLOPs (comparable across rounds, > 1 is possible) is `fp32_equivalent`, not a fraction.  split False: every layer float32, frac = that ratio."""
    t = ms * 1e-3
    fp32_eq = {"achieved": flops / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "ratio": flops / t / 1e12 / MFMA_F32_PEAK_TFLOPS,
               "note": "ALGORITHMIC float32 FLOPs / time over the float32 MFMA peak: a roofline fraction only when every layer runs on the float32 MFMA"}
    if not split:
        return {"bound": "mfma", "dtype": "f32", "achieved": fp32_eq["achieved"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp32_eq["ratio"], "fp32_equivalent": fp32_eq}
    f16_exec = 4.0 * (flops - flops_f32_layers)
    ideal = f16_exec / (MFMA_F16_PEAK_TFLOPS * 1e12) + flops_f32_layers / (MFMA_F32_PEAK_TFLOPS * 1e12)
    return {"bound": "mfma", "dtype": "f16 matrix instructions (4 executed FLOPs per algorithmic FLOP) on the layers with a split-f16 form, f32 MFMA on the rest",
            "achieved": (f16_exec + flops_f32_layers) / t / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s (executed)", "frac": ideal / t,
            "frac_note": "ideal time (executed f16 FLOPs / 2500 TFLOP/s + float32-layer FLOPs / 157.3 TFLOP/s) / measured time",
            "executed_f16_tflops": f16_exec / t / 1e12, "float32_layers_share_of_algorithmic_flops": flops_f32_layers / flops, "fp32_equivalent": fp32_eq}


def library_sha16():
    """First 16 hex digits of the sha256 of the HIP library this process runs (stamps the PMC files and the bench line)."""
    from casmvsnet_pl_amd import _lib
    try:
        with open(_lib.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def source_sha16():
    """sha256[:16] of the sources + flags the running library was compiled from (casmvsnet_pl_amd/build.py)."""
    from casmvsnet_pl_amd.build import source_sha16 as f
    return f()


def pmc_traffic(kernel_prefix, batch):
    """HBM-side bytes per launch of one kernel from the newest committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    need a pass each and cannot be collected inside a timed run; `tools/gpu_run.sh <tag> pmc` collects them over the torch-free step runner
    on the same build of the library): mean over the kernel's launches, read bytes corrected x2 as
    MI355X_MICROARCH.md prescribes.  -> (bytes | None, note, source) - `source` says which file, when it was collected
    and whether the library that produced it is the one running now (`same_library`)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if batch is None or not files:
        return None, "PMC passes are collected on the default config only", None
    path = files[-1]
    doc = json.load(open(path))
    meta = {}
    if isinstance(doc, dict):   # round 3 layout: {"meta": {...}, "kernels": [...]}
        meta, doc = doc.get("meta", {}), doc.get("kernels", [])
    if (meta.get("batch") or 2) != batch:
        return None, f"the PMC passes of {os.path.relpath(path, ROOT)} were collected at batch {meta.get('batch') or 2}, this run uses batch {batch}", None
    prefixes = (kernel_prefix,) if isinstance(kernel_prefix, str) else tuple(kernel_prefix)
    rows = [r for r in doc if r["kernel"].startswith(prefixes)]
    # the same kernels = the same sources and flags (hipcc's binaries are not bit-reproducible: the .so hash only says "the same file")
    same = (meta.get("source_sha16") == source_sha16()) if meta.get("source_sha16") else ((meta.get("library_sha16") == library_sha16()) if meta.get("library_sha16") else None)
    source = {"file": os.path.relpath(path, ROOT), "collected": meta.get("collected"), "library_sha16": meta.get("library_sha16"),
              "source_sha16": meta.get("source_sha16"), "same_library": same}
    if not rows:
        return None, "kernel not in " + os.path.relpath(path, ROOT), source
    if not source["same_library"]:
        return None, (f"{os.path.relpath(path, ROOT)} was collected on other kernel sources (source sha256 {meta.get('source_sha16')}, this run "
                      f"{source_sha16()}): not reported"), source
    n = sum(r["launches"] for r in rows)
    mb = sum((r["read_mb_corrected"] + r["write_mb"]) * r["launches"] for r in rows) / n
    return mb * 1e6, f"bytes per launch (read + write, mean over the 3 cascade levels) from {os.path.relpath(path, ROOT)}", source


def aggregate(elapsed_local, maps_local, dist=None, device=None, sum_maps=True):
    """The contract's cross-rank reduction: wall time = MAX over ranks, depth maps = SUM over ranks (replica mode: every
    rank produced its own) or the local count (view-sharded: all ranks worked on the same maps).  `dist`: an initialised
    torch.distributed module or None.  -> (elapsed, maps)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_local), int(maps_local)
    te = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    tm = torch.tensor([maps_local], dtype=torch.int64, device=device)
    if sum_maps:
        dist.all_reduce(tm, op=dist.ReduceOp.SUM)
    return float(te.item()), int(tm.item())


def base_line(metric, unit, value, world, steps, warmup, elapsed, scaling, config, median_ms=None, dtype="f32"):
    """The driver's JSON contract (one line, rank 0)."""
    line = {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic", "config": config}
    if median_ms is not None:
        line["median_ms_per_step"] = median_ms
    return line


# ---- the line the driver parses -------------------------------------------------------------------------------------------------------
# Round 5's stdout line had grown to 21.5 kB and the driver could not parse it (BENCH_r05.json: parsed null).  stdout now carries ONE line
# of at most COMPACT_LIMIT bytes: the contract's keys, `config` (short values), `roofline`, `cpu_baseline` and scalars.  The FULL object
# (stage table, batch-1 copies of every roofline object, notes) goes to --full-out (default gpurun_out/bench_full.json) and to stderr.
COMPACT_LIMIT = 4096
_CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def _r(x, digits=5):
    """Floats at `digits` significant digits (what the line is read for), everything else as is."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{digits}g}")


def _get(obj, *path):
    for k in path:
        if not isinstance(obj, dict) or k not in obj:
            return None
        obj = obj[k]
    return obj


def compact_line(full):
    """The <= 4 kB stdout line from the full result object: contract keys first, short `config`, `roofline` (dominant kernel) and `cpu_baseline`
    as objects, every other figure worth surfacing as a top-level scalar.  No prose: the notes live in DESIGN.md section 5."""
    line = {k: _r(full[k]) for k in _CONTRACT_KEYS if k in full}
    line["dtype"] = str(full.get("dtype", "f32"))[:120]
    cfg = full.get("config", {})
    short = {}
    for k in ("workload", "H", "W", "views", "n_depths", "num_groups", "batch_per_forward", "concurrent_forwards_per_gpu", "depth_interval", "init_depth_min",
              "launch", "parallelism", "arithmetic", "batch_per_gpu", "zero_grad"):
        if k in cfg:
            v = cfg[k]
            short[k] = v[:48] if isinstance(v, str) else _r(v)
    line["config"] = short
    if "median_ms_per_step" in full:
        line["median_ms_per_step"] = _r(full["median_ms_per_step"])
    rf = full.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {"kernel": str(rf.get("kernel_short") or rf.get("kernel", ""))[:80], "bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"),
                            "unit": rf.get("unit"), "frac": _r(rf.get("frac")), "traffic": _r(rf.get("traffic"), 6),
                            "traffic_over_algorithmic": _r(rf.get("traffic_over_algorithmic")), "avg_launch_ms": _r(rf.get("avg_launch_ms")),
                            "algorithmic_bytes_per_launch": _r(rf.get("algorithmic_bytes_per_launch"), 6), "launches_per_step": rf.get("launches_per_step"),
                            "mfma_executed_frac": _r(_get(rf, "mfma", "executed", "frac"))}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "median_s": _r(cb.get("median_s")), "sample": str(cb.get("sample_short") or cb.get("sample", ""))[:100]}
    scalars = {
        "roofline_homo_warp_frac": _get(full, "roofline_homo_warp", "frac"),
        "roofline_homo_warp_frac_hot": _get(full, "roofline_homo_warp", "frac_by_measurement", "reference_signature_hot"),
        "roofline_homo_warp_frac_level0": _get(full, "roofline_homo_warp", "per_level_frac", "reference_signature_dirty", "0"),
        "roofline_costvol_frac": _get(full, "roofline_costvol", "frac"),
        "roofline_costvol_traffic": _get(full, "roofline_costvol", "traffic"),
        "roofline_costreg_frac_executed": _get(full, "roofline_costreg", "frac"),
        "roofline_costreg_frac_all_float32": _get(full, "roofline_costreg", "all_float32", "frac"),
        "roofline_costreg_fp32_equivalent_ratio": _get(full, "roofline_costreg", "fp32_equivalent", "ratio"),
        "roofline_prob_regress_frac": _get(full, "roofline_prob_regress", "frac"),
        "roofline_feature_frac_executed": _get(full, "roofline_feature", "frac"),
        "roofline_feature_frac_all_float32": _get(full, "roofline_feature", "all_float32", "frac"),
        "costvol_ms_per_step": None, "costreg_ms_per_step": _get(full, "roofline_costreg", "ms_per_step"),
        "feature_ms_per_step": _get(full, "roofline_feature", "ms_per_step"),
        "tail_ms_per_step": _get(full, "roofline_prob_regress", "ms_per_step"),
        "conv3_to_conv9_ms_per_step": _get(full, "roofline_costreg", "conv3_to_conv9_ms_per_step"),
        "instrumented_ms_per_step": _get(full, "instrumented_pass", "ms_per_step"),
        "two_streams_value": _get(full, "two_streams", "value"),
        "single_stream_value": _get(full, "single_stream", "value"),
        "batch1_value": _get(full, "batch1", "value"),
        "batch1_ms_per_step": _get(full, "batch1", "ms_per_step"),
        "batch1_two_streams_value": _get(full, "batch1", "two_streams", "value"),
        "batch1_roofline_frac": _get(full, "batch1", "roofline", "frac"),
        "batch1_roofline_homo_warp_frac": _get(full, "batch1", "roofline_homo_warp", "frac"),
        "batch1_roofline_costvol_frac": _get(full, "batch1", "roofline_costvol", "frac"),
        "batch1_roofline_costreg_frac_executed": _get(full, "batch1", "roofline_costreg", "frac"),
        "train_step_ms": _get(full, "train_step", "train_step_ms"),
        "train_samples_per_s": _get(full, "train_step", "samples_per_s"),
        "train_peak_memory_gib": _get(full, "train_step", "peak_memory_gib"),
        "stock_pytorch_rocm_value": _get(full, "stock_pytorch_rocm", "value"),
        "peak_memory_gib": full.get("peak_memory_gib"),
        "library_sha16": full.get("library_sha16"), "source_sha16": full.get("source_sha16"),
    }
    st = full.get("stage_ms_per_step")
    if isinstance(st, dict):
        cv = [st.get(f"costvol_{l}") for l in range(3)]
        if all(v is not None for v in cv):
            scalars["costvol_ms_per_step"] = sum(cv)
    for m in _get(full, "conv0_other_modes", "modes") or []:
        if m.get("conv0_mode") == "f32":
            scalars["all_float32_value"] = m.get("value")
    for k in ("train_step_ms", "train_samples_per_s"):   # --mode train prints these at top level
        if scalars.get(k) is None and k in full:
            scalars[k] = full[k]
    if _get(full, "train_step", "error"):
        scalars["train_step_error"] = str(full["train_step"]["error"])[:120]
    for k, v in scalars.items():
        if v is not None:
            line[k] = _r(v)
    # last resort (a future field grows): drop scalars from the back until the line fits; the contract keys, roofline and cpu_baseline stay
    keys = [k for k in line if k not in _CONTRACT_KEYS and k not in ("config", "roofline", "cpu_baseline")]
    while len(json.dumps(line)) > COMPACT_LIMIT and keys:
        line.pop(keys.pop())
    return line


def emit(full, full_out):
    """stdout: the compact line (<= COMPACT_LIMIT bytes).  The full object: `full_out` (best effort: a read-only tree must not lose the line) + stderr."""
    text = json.dumps(full)
    if full_out:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_out)), exist_ok=True)
            with open(full_out, "w") as f:
                f.write(text + "\n")
        except OSError as e:
            print(f"warning: could not write {full_out}: {e}", file=sys.stderr)
    print("bench_full: " + text, file=sys.stderr, flush=True)
    line = compact_line(full)
    out = json.dumps(line)
    assert len(out) <= COMPACT_LIMIT, len(out)
    print(out, flush=True)


def cpu_baseline(cfg_name):
    """The reference's forward on the same synthetic workload on this host's cores, at the best of a sweep over the thread
    count (all 256 hardware threads of the GPU box are 25x slower than 8: oversubscription).  With /root/reference
    present (the build container) it is the UNMODIFIED reference module behind the two import shims (kind "reference");
    on the GPU box, where that tree does not exist, the oracle's restatement of it (kind "port")."""
    from oracle import cpu_restatement as R
    from oracle import reference_loader as RL
    H, W, V, G, n_depths, ratios, _ = CONFIGS[cfg_name]
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    sd = randomize_state_dict(model.state_dict(), seed=0)
    imgs, proj, dmin, dint = config_inputs(cfg_name, 1, seed=0)
    if RL.reference_available():
        kind = "reference"
        ref = RL.build_reference_model(n_depths, ratios, G, sd)

        def forward():
            with torch.no_grad():
                return ref(imgs, proj, dmin, dint)
    else:
        kind = "port"

        def forward():
            return R.cascade_forward(sd, imgs, proj, dmin, dint, n_depths, ratios, G)
    ncpu = os.cpu_count() or 8
    old = torch.get_num_threads()
    sweep = sorted({t for t in (4, 8, 16, 32, 64) if t <= ncpu})   # beyond 64 threads torch's CPU ops only lose (256 threads: 54 s per forward)
    best, per_threads = None, {}
    t_start = time.perf_counter()
    for i, nt in enumerate(sweep):
        torch.set_num_threads(nt)
        if i == 0:
            forward()  # warm-up (allocator, op dispatch)
        t0 = time.perf_counter()
        forward()
        per_threads[nt] = time.perf_counter() - t0
        if best is None or per_threads[nt] < per_threads[best]:
            best = nt
        if time.perf_counter() - t_start > 40.0:
            break
    torch.set_num_threads(best)
    times = [per_threads[best]]
    for _ in range(2):
        t0 = time.perf_counter()
        forward()
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(old)
    times.sort()
    med = times[len(times) // 2]
    what = "the unmodified /root/reference models/mvsnet.py (import shims: inplace_abn, kornia)" if kind == "reference" else \
           "oracle/cpu_restatement.py (the reference tree is not on this machine)"
    return {"value": 1.0 / med, "unit": "depth-maps/s", "cores": best, "kind": kind, "median_s": med,
            "sample_short": f"3 forwards of 1 depth map, {cfg_name}, torch CPU fp32, best of a thread sweep",
            "sample": f"3 timed forwards of ONE depth map each (median {med:.3f} s) of {what} on the same {cfg_name} inputs / weights, "
                      f"torch CPU fp32 at {best} threads = the best of a sweep {({k: round(v, 2) for k, v in per_threads.items()})} s over "
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
    """EXACTLY `steps` steps between two barrier + synchronize pairs; one HIP event per step (recorded on the launch
    stream, no synchronisation) gives the per-step durations for the median."""
    events = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for e in events:
        e.record()   # creates the hipEvent_t outside the timed region
    barrier()
    t0 = time.perf_counter()
    events[0].record()
    for i in range(steps):
        out = step()
        events[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(events[i].elapsed_time(events[i + 1]) for i in range(steps))
    median = per_step[len(per_step) // 2] if steps % 2 else 0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2])
    return elapsed, out, median


_DIRTY = {}


def _timed_op(fn, reps, dirty_mb, dev):
    """Mean ms of one call of fn: `reps` back-to-back calls between two events (hot: inputs stay in L2 / Infinity Cache),
    or - dirty_mb > 0 - each call timed on its own after a torch fill_ of that many MB (ordinary stores): the caches then
    hold another kernel's dirty lines and none of the inputs, the state every kernel starts from inside the forward."""
    for _ in range(3):
        fn()
    if not dirty_mb:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    buf = _DIRTY.get(dirty_mb)
    if buf is None:
        buf = _DIRTY[dirty_mb] = torch.empty(dirty_mb * 262144, device=dev)
    total = 0.0
    for _ in range(reps):
        buf.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        total += s.elapsed_time(e)
    return total / reps


def homo_warp_roofline(dev, H, W, n_depths, B):
    """The un-fused op models/modules.py:52-92 (north_star names its HBM-roofline fraction) at the three level shapes,
    algorithmic bytes 4 B (C h w + D h w + C D h w), measured four ways:
      reference signature (NCHW source in: ops.homo_warp - since round 4 the box is staged from the channel planes, no layout pass) and the
      kernel on a pixel-major source (what FeatureNet hands the engine), each hot (10 back-to-back calls) and with dirtied caches
      (a 512 MB fill_ between calls = the state inside the forward).  `frac` is the most conservative of the four:
      reference signature, dirtied caches."""
    from casmvsnet_pl_amd import ops
    from casmvsnet_pl_amd.synthetic import make_inputs
    _, proj, dmin, dint = make_inputs(B, 2, H, W, seed=0)
    work = algorithmic_work(H, W, 2, 1, n_depths, B)
    variants = {"reference_signature_dirty": (True, 512), "reference_signature_hot": (True, 0),
                "pixel_major_kernel_dirty": (False, 512), "pixel_major_kernel_hot": (False, 0)}
    tot_b = sum(work[l]["homo_warp_bytes"] for l in range(3))
    ms = {k: {} for k in variants}
    for l in range(3):
        C, D, h, w = 8 * 2 ** l, n_depths[l], H >> l, W >> l
        nchw = torch.randn(B, C, h, w, device=dev)
        nhwc = nchw.permute(0, 2, 3, 1).contiguous()   # pixel-major, as FeatureNet hands it to the engine
        P = proj[:, 0, l].contiguous().to(dev)
        step = dint * 2 ** l
        k = torch.arange(D, device=dev, dtype=torch.float32).view(1, D, 1, 1)
        depth = (680.0 - D / 2 * step + 40.0 * torch.sin(torch.linspace(0, 6.0, w, device=dev)).view(1, 1, 1, w) + k * step).expand(B, D, h, w).contiguous()
        for name, (ref_sig, dirty) in variants.items():
            fn = (lambda: ops.homo_warp(nchw, P, depth)) if ref_sig else (lambda: ops.homo_warp_nhwc(nhwc, P, depth))
            ms[name][l] = _timed_op(fn, 10, dirty, dev)
    fr = {name: tot_b / (sum(v.values()) * 1e-3) / 1e9 / HBM_PEAK_GBS for name, v in ms.items()}
    per_level = {name: {str(l): work[l]["homo_warp_bytes"] / (v[l] * 1e-3) / 1e9 / HBM_PEAK_GBS for l in range(3)} for name, v in ms.items()}
    head = "reference_signature_dirty"
    return {"kernel": "homo_warp (un-fused op modules.py:52-92): costvol_lds_kernel<MODE_WARP_NCHW> through the reference signature "
                      "(ops.homo_warp: the source box staged straight from the (B, C, H, W) map), one call per level shape, caches dirtied between calls",
            "bound": "hbm", "achieved": fr[head] * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr[head], "traffic": None,
            "frac_by_measurement": fr, "per_level_frac": per_level, "avg_launch_ms": sum(ms[head].values()) / 3}


def instrumented_pass(model, inputs, cfg_name, B, K, every, barrier, dev, with_homo_warp=True):
    """K kernel-by-kernel steps with HIP events around every kernel on every `every`-th one -> roofline objects + stage times."""
    H, W, V, G, n_depths, ratios, _ = CONFIGS[cfg_name]
    n_ev = (K + every - 1) // every
    timer = StageTimer()
    timer.reserve((2 * 13 + 14 + 36) * n_ev)
    for _ in range(2):   # untimed: the eager pass allocates outside the captured graph's pool - at 1152x864 x 5 views the first kernel-by-kernel step
        model(*inputs)   # spends tens of ms in hipMalloc between the events of a stage (costvol_1 read 22.9 ms for a 4 ms kernel)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):   # events on every `every`-th step: the GPU stays busy in between
        model.set_timer(timer if i % every == 0 else None)
        model(*inputs)
    barrier()
    eager_ms = 1e3 * (time.perf_counter() - t0) / K
    model.set_timer(None)
    summ = timer.summary(LAYER_NAMES)
    work = algorithmic_work(H, W, V, G, n_depths, B)
    per_step = {k: v["ms"] / n_ev for k, v in summ.items()}
    out = {}
    # dominant kernel: CostRegNet.conv0 (3 launches per step).  What binds it: with its products on the f16 matrix cores the layer's
    # HBM time (input + output volume once, 8 TB/s) and its matrix time (3 partial products x 4/3 K padding at 2.5 PFLOP/s) are about equal
    # (0.25 / 0.23 ms at level 1, batch 8) - the larger one, HBM, is the roof the fraction is quoted against; the matrix view sits beside it.
    conv0_ms = sum(summ[f"costreg_{l}/conv0"]["ms"] for l in range(3))
    conv0_flops = sum(work[l]["conv0_flops"] for l in range(3)) * n_ev
    ach = conv0_flops / (conv0_ms * 1e-3) / 1e12
    mode = getattr(model.cost_reg_0, "_conv0_active", None) or "f32"
    split = mode if mode in ("splitbf16", "splitf16") and getattr(model, "fuse_regress", False) and G in (1, 8) else None
    knames = {"splitbf16": ("conv0_sb_kernel",), "splitf16": ("conv0_sf_kernel", "conv0_zm_kernel", "conv0_zw_kernel"), None: ("conv16db_kernel<2, 4, 4, 4, 4, 32",)}[split]
    traffic, traffic_note, src = pmc_traffic(knames, B if cfg_name == HEADLINE else None)
    conv0_alg = sum(4 * B * ((G if G > 1 else 8 * 2 ** l) + 8) * n_depths[l] * (H >> l) * (W >> l) for l in range(3))   # bytes per step: input + output volumes
    gbs = conv0_alg * n_ev / (conv0_ms * 1e-3) / 1e9
    avg_launch_s = conv0_ms * 1e-3 / (3 * n_ev)
    out["roofline"] = {"kernel": {"splitbf16": "conv0_sb_kernel<CIN, 6> (CostRegNet.conv0 on the bf16 matrix cores, float32 operands as three exact "
                                               "bf16 slices; 3 launches per step)",
                                  "splitf16": "CostRegNet.conv0 on the f16 matrix cores (float32 operands as two float16 slices behind exact power-of-two "
                                              "scalings), 3 launches per step: conv0_zw_kernel<8, wide> (level 0), <16, wide> (level 1), <32> (level 2): "
                                              "input-stationary along z on 8 x 64 / 16 x 32 patches, producer and consumer wave groups in one workgroup",
                                  None: "conv16db_kernel<PX> (CostRegNet.conv0 on the float32 MFMA: Cout 8, stride 1; 3 launches per step)"}[split],
                       "kernel_short": {"splitbf16": "conv0_sb_kernel<CIN,6> (CostRegNet.conv0, bf16 slices)", "splitf16": "conv0_zw_kernel<CIN,WIDE> (CostRegNet.conv0, split-f16)",
                                        None: "conv16db_kernel<PX> (CostRegNet.conv0, f32 MFMA)"}[split],
                       "launches_per_step": 3, "bound": "hbm" if split else "mfma", "batch": B, "avg_launch_ms": conv0_ms / (3 * n_ev)}
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": conv0_alg / 3,
           "note": "achieved = ALGORITHMIC bytes (each launch reads its input volume and writes its 8-channel output once; mean of the 3 levels) / HIP-event time"}
    mfma_view = {"fp32_equivalent": {"achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "ratio": ach / MFMA_F32_PEAK_TFLOPS,
                                     "note": "the layer's ALGORITHMIC float32 FLOPs (2 * 27 * cin * 8 per voxel) / time over the float32 MFMA peak: comparable "
                                             "across rounds; a roofline fraction only when conv0 runs on the float32 MFMA (conv0_other_modes)"}}
    if split:   # what the matrix cores actually execute: 6 bf16 (3 f16) products per float32 product, 4 K-slots per 3 taps
        mult = {"splitbf16": 8.0, "splitf16": 4.0}[split]
        mfma_view["executed"] = {"dtype": {"splitbf16": "bf16", "splitf16": "f16"}[split], "flops_per_algorithmic_flop": mult, "achieved": mult * ach,
                                 "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mult * ach / MFMA_F16_PEAK_TFLOPS,
                                 "note": "the dense bf16 and f16 matrix peaks are equal (2.5 PFLOP/s)"}
        out["roofline"].update(hbm)
    else:
        out["roofline"].update({"achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS, "hbm": hbm})
    out["roofline"]["mfma"] = mfma_view
    out["roofline"]["traffic"] = traffic
    out["roofline"]["traffic_note"] = traffic_note
    out["roofline"]["traffic_source"] = src
    if traffic is not None:
        out["roofline"]["traffic_frac"] = traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS     # counter bytes / time over the HBM peak
        out["roofline"]["traffic_over_algorithmic"] = traffic / (conv0_alg / 3)
    cr_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("costreg_"))
    cr_flops = sum(work[l]["costreg_flops"] for l in range(3)) * n_ev
    fused = getattr(model, "fuse_regress", False)
    cr_split = getattr(model.cost_reg_0, "ci_mode", "f32") == "splitf16" and split is not None
    cr_f32_flops = sum(COSTREG_F32_FLOPS_PER_VOXEL * n_depths[l] * (H >> l) * (W >> l) * B for l in range(3)) * n_ev
    out["roofline_costreg"] = {"kernel": "all 33 CostRegNet launches" + (" (the `prob` interval includes the fused softmax regression)" if fused else ""),
                               **mixed_mfma_roofline(cr_flops, cr_f32_flops, cr_ms, cr_split), "ms_per_depth_map": cr_ms / n_ev / B, "ms_per_step": cr_ms / n_ev, "batch": B}
    rest_ms = cr_ms - conv0_ms
    out["roofline_costreg"]["without_conv0"] = {"ms_per_step": rest_ms / n_ev, **{k: v for k, v in mixed_mfma_roofline(cr_flops - conv0_flops, cr_f32_flops, rest_ms, cr_split).items()
                                                                                    if k in ("frac", "achieved", "fp32_equivalent")}}
    out["roofline_costreg"]["conv3_to_conv9_ms_per_step"] = sum(per_step[f"costreg_{l}/{n}"] for l in range(3) for n in ("conv3", "conv4", "conv5", "conv6", "conv7", "conv9"))
    cv_ms = sum(summ[f"costvol_{l}"]["ms"] for l in range(3))
    cv_bytes = sum(work[l]["costvol_bytes"] for l in range(3)) * n_ev
    cv_traffic, cv_note, cv_src = pmc_traffic(("costvol_lds_kernel", "costvol_nhwc_kernel"), B if cfg_name == HEADLINE else None)
    from casmvsnet_pl_amd import _lib
    lds = [bool(_lib.load().casmvs_costvol_lds_preferred(8 * 2 ** l, W >> l, n_depths[l], V - 1, G)) for l in range(3)]
    out["roofline_costvol"] = {"kernel": "fused homo_warp + aggregation, one launch per level: " +
                                         ", ".join(f"level {l}: {'costvol_lds_kernel' if lds[l] else 'costvol_nhwc_kernel'}<C={8 * 2 ** l}>" for l in (2, 1, 0)),
                               "bound": "hbm", "achieved": cv_bytes / (cv_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": cv_bytes / (cv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "traffic": cv_traffic, "traffic_note": cv_note + f"; algorithmic: {cv_bytes / n_ev / 3:.4g}", "traffic_source": cv_src,
                               "ms_per_depth_map": cv_ms / n_ev / B, "batch": B,
                               "per_level_frac": {str(l): work[l]["costvol_bytes"] * n_ev / (summ[f"costvol_{l}"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS for l in range(3)}}
    if fused:
        # the regulariser's tail = conv11 (+ skip) + `prob` + regression: ONE depth-walking kernel where its tiles fill the chip (conv11_prob_zfused.hip; the
        # `conv11` interval is then empty and `prob` times the fused kernel), else deconv11_sf_kernel + prob_zwalk_kernel.  Algorithmic bytes of the pair WITHOUT
        # the 8-channel tensor between them (conv9's output: 16 channels at n / 8 voxels = 2 n floats, the skip tensor 8 n, hypotheses n, cost n, two maps): what the
        # fused kernel has to move; the two-kernel form moves 16 n more.
        pr_ms = max(1e-6, sum(summ[f"costreg_{l}/prob"]["ms"] + summ[f"costreg_{l}/conv11"]["ms"] for l in range(3)))
        pr_bytes = sum(work[l]["tail_bytes"] for l in range(3)) * n_ev
        zf = [B * -(-(H >> l) // 16) * -(-(W >> l) // 60) >= 180 and getattr(getattr(model, f"cost_reg_{l}"), "_ci_active", False) for l in range(3)]
        out["roofline_prob_regress"] = {"kernel": "CostRegNet's tail (mvsnet.py:84-89,101,104,174-193), per level: " +
                                                  ", ".join(f"level {l}: " + ("conv11_prob_zfused_kernel" if zf[l] else "deconv11_sf_kernel + prob_zwalk_kernel (+ softmax_regress_kernel)")
                                                            for l in (2, 1, 0)), "bound": "hbm",
                                        "achieved": pr_bytes / (pr_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": pr_bytes / (pr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_step": pr_ms / n_ev, "batch": B,
                                        "note": "algorithmic bytes = conv9's output + the skip tensor + hypotheses + cost + depth / confidence; the tensor between conv11 and "
                                                "`prob` (16 n floats of traffic in the two-kernel form) is not counted"}
    else:
        sm_ms = sum(summ[f"softmax_{l}"]["ms"] for l in range(3))
        sm_bytes = sum(work[l]["softmax_bytes"] for l in range(3)) * n_ev
        out["roofline_softmax"] = {"kernel": "softmax_regress_kernel (3 launches)", "bound": "hbm",
                                   "achieved": sm_bytes / (sm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": sm_bytes / (sm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "batch": B}
    ft_ms = sum(v["ms"] for k, v in summ.items() if k.startswith("feature/"))
    if ft_ms > 0:
        ft_flops = feature_flops(H, W) * V * B * n_ev
        ft_split = getattr(model.feature, "tail_mode", "f32") == "splitf16" and getattr(model.feature, "fuse_tail", False)
        out["roofline_feature"] = {"kernel": "all FeatureNet launches", **mixed_mfma_roofline(ft_flops, feature_flops_f32_layers(H, W, bool(getattr(model.feature, "fuse_conv0", False))) * V * B * n_ev, ft_ms, ft_split),
                                   "ms_per_depth_map": ft_ms / n_ev / B, "ms_per_step": ft_ms / n_ev, "batch": B}
    if with_homo_warp:
        out["roofline_homo_warp"] = homo_warp_roofline(dev, H, W, n_depths, B)
        out["roofline_homo_warp"]["batch"] = B
    out["stage_ms_per_step"] = {k: round(v, 4) for k, v in per_step.items()}
    out["instrumented_pass"] = {"steps": K, "event_sampled_steps": n_ev, "ms_per_step": eager_ms,
                                "note": f"kernel-by-kernel launches right after the timed steps, ~90 HIP events on every {every}-th step"}
    return out


def train_mode(args, dev, world, rank, dist, barrier):
    """--mode train: the reference's training step (train.py:99-127) through the HIP training path - train-mode forward
    (batch-statistics InPlaceABN), SL1 loss over the three levels (losses.py, masked), backward, SGD update -, captured as
    ONE hipGraph like the inference forward (--no-graph: eager).  metric: training samples/s."""
    from casmvsnet_pl_amd import InPlaceABN
    from casmvsnet_pl_amd import training
    from casmvsnet_pl_amd.training import sl1_loss, sl1_loss_masked
    H, W, V, G, n_depths, ratios, _ = CONFIGS[args.config]
    B = args.batch if args.batch_given else 1   # the reference's default: --batch_size 1
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=InPlaceABN)
    randomize_state_dict(model.state_dict(), seed=0)
    model = model.to(dev).train()
    net = model
    if dist is not None and world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])
    imgs, proj, dmin, dint = config_inputs(args.config, B, seed=rank)
    imgs, proj = imgs.to(dev), proj.to(dev)
    g = torch.Generator().manual_seed(100 + rank)
    depths = {f"level_{l}": (dmin + dint * 96 * torch.rand(B, H >> l, W >> l, generator=g)).to(dev) for l in range(3)}
    masks = {f"level_{l}": (torch.rand(B, H >> l, W >> l, generator=g) > 0.2).to(dev) for l in range(3)}
    opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
    use_graph = not args.no_graph and world == 1
    # losses.py indexes with the boolean mask (a host sync per level): kept kernel by kernel; the captured step uses the
    # same loss in its sync-free form (mean over the masked elements through a float mask)
    loss_fn = sl1_loss_masked if use_graph else sl1_loss

    def step(zero=True):
        # gradients dropped, not zero-filled (torch's default since 2.0): a zero-filled gradient makes autograd ADD into it - one more launch per
        # parameter and step (115 `add` launches = 0.7 ms of the kernel-by-kernel step in profiles/r02_s3_train_step_kernel_stats.csv).
        # --zero-fill-grads: the behaviour before round 3's last session (A/B).
        if zero:
            opt.zero_grad(set_to_none=not args.zero_fill_grads)
        loss = loss_fn(net(imgs, proj, dmin, dint), depths, masks)
        loss.backward()
        opt.step()
        return loss

    run = step
    if use_graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(3, args.warmup)):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        from casmvsnet_pl_amd import streams
        streams.reset(dev)   # the device is idle: whatever ran before on other streams (the inference legs of the default line) cannot overlap the capture

        def capture(zero_fill):
            graph = torch.cuda.CUDAGraph()
            if not zero_fill:
                # torch's whole-network capture pattern: no gradients exist when the capture starts, so the captured backward allocates them from the
                # graph's pool and WRITES them (no accumulate-add); every replay refills the same tensors, zero_grad is not part of the step
                opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                loss = step(zero=zero_fill)
            return graph, loss
        try:
            graph, static_loss = capture(args.zero_fill_grads)
        except RuntimeError as e:   # (the dropped-gradient capture was written without a GPU run: keep the measured form as the fallback)
            if args.zero_fill_grads:
                raise
            print(f"warning: capture with dropped gradients failed ({str(e).splitlines()[0][:200]}); falling back to zero-filled gradients", file=sys.stderr)
            torch.cuda.synchronize()
            args.zero_fill_grads = True
            step()
            graph, static_loss = capture(True)

        def run():
            graph.replay()
            return static_loss
        for _ in range(2):
            run()
    else:
        for _ in range(max(3, args.warmup)):
            step()
    elapsed, loss, median = timed_steps(run, args.steps, barrier)
    assert torch.isfinite(loss).all()
    elapsed, samples = aggregate(elapsed, B * args.steps, dist if world > 1 else None, dev)
    if rank != 0:
        return None
    line = base_line(f"training samples/sec ({args.config}: train-mode forward + SL1 loss + backward + SGD, train.py:99-127)", "samples/s",
                     samples / elapsed, world, args.steps, args.warmup, elapsed, "weak",
                     {"workload": args.config + "_train", "H": H, "W": W, "views": V, "n_depths": list(n_depths), "num_groups": G,
                      "batch_per_gpu": B, "launch": "one hipGraph replay per step" if use_graph else "kernel by kernel",
                      "zero_grad": "zero-filled gradients + accumulate-adds" if args.zero_fill_grads else "set_to_none",
                      "parallelism": f"DistributedDataParallel x{world} over RCCL" if world > 1 else "single GPU"}, median)
    line["train_step_ms"] = 1e3 * elapsed / args.steps
    line["peak_memory_gib"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    return line


def torchrun_command(n, argv, port=None):
    """The contract's launch line for N ranks on one node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or os.environ.get("MASTER_PORT", 29531)), os.path.abspath(__file__)] + list(argv)


def relaunch_under_torchrun(n, argv):
    import subprocess
    avail = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if avail < n:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {avail} GPU(s)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(torchrun_command(n, argv), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=HEADLINE, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None,
                    help="depth maps per forward per GPU (default 8: independent reference views batched like the reference's train.py "
                         "--batch_size; --batch 1 = the reference's eval.py loop, always measured too).  --mode train: default 1")
    ap.add_argument("--mode", default="replica", choices=["replica", "view_sharded", "train"])
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying one hipGraph")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent forwards in flight per GPU, one HIP stream + hipGraph each (a step = one round of all of them; --batch is per "
                         "forward).  The replicas keep the model's split-f16 layers: the library carries no packed-float32 instruction of the class that "
                         "is wrong beside f16 matrix instructions on the MI355X (casmvsnet_pl_amd/streams.py).  With the default single stream, 2 "
                         "streams x half the batch and 2 streams x batch 1 are measured beside the headline (`two_streams`, `batch1.two_streams`).")
    ap.add_argument("--float32-replicas", action="store_true",
                    help="with --streams > 1: every layer of the replicas on the float32 MFMA kernels (the round 2-4 configuration)")
    ap.add_argument("--unsafe-mixed-streams", action="store_true", help="(obsolete: the replicas keep the split-f16 layers by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stock-pytorch", action="store_true",
                    help="also time the restated reference forward on stock PyTorch-ROCm operators on this GPU (adds ~1 min of MIOpen search)")
    ap.add_argument("--no-events", action="store_true", help="skip the instrumented pass (no roofline objects)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="the instrumented pass records its ~90 HIP events on every n-th of its K kernel-by-kernel steps (an event "
                         "costs ~3 us of GPU time; on every step they would inflate the step by ~10 %% and starve the launch queue)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the extra batch-1 measurement")
    ap.add_argument("--no-fuse-regress", action="store_true", help="A/B: `prob` and the softmax regression as separate library calls")
    ap.add_argument("--conv0-mode", default=None, choices=["splitf16", "splitbf16", "f32"],
                    help="CostRegNet.conv0: 'splitf16' = float32 operands as two scaled float16 slices on the f16 matrix cores (the model's default), "
                         "'splitbf16' = three exact bf16 slices on the bf16 matrix cores, 'f32' = the float32 MFMA kernel like every other layer; "
                         "'splitf16' also runs conv2 / conv4 in that arithmetic, the other two keep them in float32; the other modes' throughputs "
                         "are measured and printed beside the headline")
    ap.add_argument("--fuse-tail", type=int, default=None, help="A/B: FeatureNet's full-resolution FPN tail as one kernel (1) or as the reference's three steps (0); default: the model's")
    ap.add_argument("--zero-fill-grads", action="store_true", help="--mode train A/B: optimizer.zero_grad(set_to_none=False) as before round 3's last session")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where the FULL result object goes (stage table, batch-1 roofline copies, notes); stdout carries the compact line only; '' = nowhere")
    ap.add_argument("--no-train-step", action="store_true", help="skip the `train_step` object of the default line (20 hipGraph replays of the batch-1 training step)")
    args = ap.parse_args()
    args.batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 8

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL, the contract's launch line)
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
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

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        line = train_mode(args, dev, world, rank, dist, barrier)
        if rank == 0:
            line["library_sha16"] = library_sha16()
            emit(line, args.full_out)
        if dist is not None:
            dist.destroy_process_group()
        return

    H, W, V, G, n_depths, ratios, _ = CONFIGS[args.config]
    view_sharded = args.mode == "view_sharded"

    conv0_mode = [None]   # override for the "other mode" measurement below

    def build(B):
        model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
        randomize_state_dict(model.state_dict(), seed=0)
        model = model.to(dev).eval()
        model.fuse_regress = not args.no_fuse_regress
        if args.fuse_tail is not None:
            model.feature.fuse_tail = bool(args.fuse_tail)
        mode = conv0_mode[0] or args.conv0_mode
        if mode is not None:
            for l in range(3):   # "splitf16" = every layer that has an f16 form (conv0, conv2, conv4); the other two keep conv2 / conv4 in float32
                getattr(model, f"cost_reg_{l}").conv0_mode = mode
                getattr(model, f"cost_reg_{l}").ci_mode = "splitf16" if mode == "splitf16" else "f32"
            model.feature.tail_mode = "splitf16" if mode == "splitf16" else "f32"
        # replica: every rank works on its own depth maps (different seeds -> different images / cameras);
        # view_sharded: all ranks share the depth maps and split their source views
        imgs, proj, dmin, dint = config_inputs(args.config, B, seed=0 if view_sharded else rank)
        if view_sharded:
            model.view_shard_group = dist.group.WORLD
        return model, imgs.to(dev), proj.to(dev), dmin, dint

    def measure(B, steps, warmup, streams=1):
        model, imgs, proj, dmin, dint = build(B)
        use_graph = not args.no_graph and not view_sharded   # a collective inside a capture is not attempted
        if use_graph:
            model(imgs, proj, dmin, dint)   # lazy packing / workspaces before the capture (GraphedForward warms up again on its capture stream)
        if use_graph and streams > 1:
            cf = ConcurrentForwards(model, imgs, proj, dmin, dint, n_streams=streams, mixed_matrix_types=False if args.float32_replicas else None)
            if not cf.mixed_matrix_types:   # what the replicas run (and what the instrumented pass below should time)
                for l in range(3):
                    getattr(model, f"cost_reg_{l}").conv0_mode = getattr(model, f"cost_reg_{l}").ci_mode = "f32"
                model.feature.tail_mode = "f32"
            step = lambda: cf.run()[-1]
        elif use_graph:
            gf = GraphedForward(model, imgs, proj, dmin, dint)
            step = lambda: gf(imgs, proj)
        else:
            step = lambda: model(imgs, proj, dmin, dint)
        # the W untimed warm-up steps are steps of the SAME kind as the timed ones (round 6: they were eager forwards before the capture, which left the first
        # replays of the hipGraph - its upload - inside the timed region: mean 6.72 ms against a median of 6.61)
        for _ in range(warmup):
            step()
        elapsed, out, median = timed_steps(step, steps, barrier)
        maps_local = B * (streams if use_graph else 1) * steps
        elapsed, maps = aggregate(elapsed, maps_local, dist, dev, sum_maps=not view_sharded)
        assert torch.isfinite(out["depth_0"]).all()
        return model, (imgs, proj, dmin, dint), elapsed, maps, median, use_graph

    B = args.batch
    NS = 1 if (view_sharded or args.no_graph) else max(1, args.streams)
    model, inputs, elapsed, maps, median, used_graph = measure(B, args.steps, args.warmup, NS)
    K = args.steps
    line = None
    if rank == 0:
        line = base_line("depth-maps/sec at 640x512, 3 views, n_depths=[8,32,48]" if args.config == HEADLINE else f"depth-maps/sec ({args.config})",
                         "depth-maps/s", maps / elapsed, world, K, args.warmup, elapsed, "strong" if view_sharded else "weak",
                         {"workload": args.config, "H": H, "W": W, "views": V, "n_depths": list(n_depths),
                          "interval_ratios": list(ratios), "num_groups": G, "depth_maps_per_step_per_gpu": B * NS,
                          "batch_per_forward": B, "concurrent_forwards_per_gpu": NS,
                          "depth_interval": inputs[3], "init_depth_min": inputs[2],
                          "launch": (f"{NS} hipGraphs on {NS} streams per step" if NS > 1 else "one hipGraph replay per step") if used_graph else "kernel by kernel",
                          "parallelism": f"view-sharded x{world}, 1 all-reduce per level" if view_sharded else f"replica x{world}, no collective",
                          "arithmetic": "split-f16" if model.cost_reg_0.conv0_mode == "splitf16" else model.cost_reg_0.conv0_mode,
                          "parallelism_note": (f"view-sharded x{world}: source views split over the ranks, one RCCL all-reduce of the sum / "
                                               "sum-of-squares volumes per level, every rank regularises") if view_sharded else
                                              f"replica x{world} (one depth map stream per GPU, no data-path collective)",
                          "feature_net": "HIP MFMA kernels (casmvs_featurenet_forward_f32)",
                          "regression": "fused into the `prob` head's library call (casmvs_costreg_regress_f32)" if model.fuse_regress else "separate launch",
                          "conv0_arithmetic": {"splitbf16": "float32 operands as three exact bf16 slices, six bf16 x bf16 partial products per product on the "
                                                            "bf16 matrix cores, float32 accumulation (conv0_splitbf16.hip)",
                                               "splitf16": "float32 operands as two float16 slices (22 significand bits) behind exact power-of-two scalings, "
                                                           "three f16 x f16 partial products per product on the f16 matrix cores, float32 accumulation "
                                                           "(conv0_zmarch.hip: input-stationary along z, producer / consumer wave groups); "
                                                           "float32-grade: distance to a float64 convolution at or below the float32 MFMA kernel's",
                                               "f32": "float32 MFMA (v_mfma_f32_16x16x4_f32)"}[model.cost_reg_0.conv0_mode],
                          "conv1_to_conv11_arithmetic": "conv1 / conv2 / conv3 / conv4 / conv6 / conv9 / conv11 as conv0's split-f16 (conv_s2_splitf16.hip, conv_ci_splitf16.hip, "
                                                        "deconv9_splitf16.hip, deconv11_splitf16.hip); conv3 / conv4 / conv6 only where the volume gives >= 100 patches / tiles "
                                                        "(conv6: >= 3 planes); conv5 / conv7 and `prob` float32" if model.cost_reg_0.ci_mode == "splitf16" else "float32 MFMA",
                          "per_gpu_engine": f"one stream, batch {B}, one hipGraph replay per step, split-f16 layer set (what every rank of a replica run executes)"
                                            if used_graph and NS == 1 and model.cost_reg_0.conv0_mode == "splitf16" else "see launch / *_arithmetic",
                          "featurenet_arithmetic": ("fused FPN tail, conv1.1 / conv1.2 / conv2.1 / conv2.2 / smooth1 and the 5x5 stride-2 layers conv1.0 / conv2.0 as conv0's "
                                                    "split-f16 (fpn_fused_sf.hip, conv2d_ci_splitf16.hip, conv2d_k5s2_splitf16.hip), conv0.0 + conv0.1 as one f16 kernel (fnet_conv0_mm.hip); the "
                                                    "1x1 laterals float32 MFMA" if model.feature.tail_mode == "splitf16"
                                                    else "float32 MFMA (fused FPN tail: fpn_fused.hip)")
                                                   if model.feature.fuse_tail else "float32 MFMA, the FPN tail as three steps (lat0, upsample-add, smooth0)",
                          "every_layer_float32_value": "conv0_other_modes.modes[conv0_mode == 'f32'] of this line"},
                         median,
                         dtype="f32" if model.cost_reg_0.conv0_mode == "f32" and model.cost_reg_0.ci_mode == "f32" and model.feature.tail_mode == "f32" else
                               "f32 (tensors + accumulation f32; products of 19 layers on the f16 MFMA from 2 f16 slices per operand)")
        line["library_sha16"] = library_sha16()
        line["source_sha16"] = source_sha16()

    # ---- instrumented eager pass: HIP events around every kernel (same model, same inputs) ------------------------
    if not args.no_events:
        res = instrumented_pass(model, inputs, args.config, B, K, max(1, args.event_every), barrier, dev)
        if rank == 0:
            line.update(res)
            # what north_star names, as top-level scalars (they survive a parser that keeps scalars only)
            line["roofline_homo_warp_frac"] = res["roofline_homo_warp"]["frac"]                                       # reference signature, caches dirtied
            line["roofline_homo_warp_frac_hot"] = res["roofline_homo_warp"]["frac_by_measurement"]["reference_signature_hot"]
            line["roofline_costvol_frac"] = res["roofline_costvol"]["frac"]
        if (args.conv0_mode or "splitf16") != "f32" and not args.no_batch1:
            # the MEASURED float32-MFMA fraction of CostRegNet / FeatureNet (north_star: >= 0.4 MFMA utilisation on CostRegNet): the same launch with every
            # layer on the float32 MFMA kernels, kernel by kernel under HIP events - a fraction of the peak those kernels run on
            conv0_mode[0] = "f32"
            m32, in32 = build(B)[0], inputs
            for _ in range(2):
                m32(*in32)
            r32 = instrumented_pass(m32, in32, args.config, B, max(4, K // 2), max(1, args.event_every), barrier, dev, with_homo_warp=False)
            conv0_mode[0] = None
            del m32
            if rank == 0:
                for key in ("roofline_costreg", "roofline_feature"):
                    line[key]["all_float32"] = {k: r32[key][k] for k in ("frac", "achieved", "peak", "unit", "ms_per_step") if k in r32[key]}
                    line[key]["all_float32"]["note"] = "the same stage with every layer on the float32 MFMA kernels (conv0_mode = ci_mode = tail_mode = f32), measured"
                line["roofline_costreg_frac_all_float32"] = r32["roofline_costreg"]["frac"]
    del model
    # ---- one forward per step (no concurrency), and the reference's eval.py loop: one reference view per step --------
    if NS > 1:
        _, _, els, ms1, med1, _ = measure(B * NS, K, max(2, args.warmup // 2), 1)
        if rank == 0:
            line["single_stream"] = {"value": ms1 / els, "unit": "depth-maps/s", "ms_per_step": 1e3 * els / K, "median_ms_per_step": med1, "steps": K,
                                     "note": f"the same {B * NS} depth maps per step as ONE forward (one stream, one hipGraph replay)"}
    if NS == 1 and used_graph and not args.no_batch1 and B % 2 == 0:
        # the same depth maps per step as TWO concurrent forwards of half the batch (graph.ConcurrentForwards: the split-f16 layers on both streams)
        _, _, el2, ms2, med2, _ = measure(B // 2, K, max(2, args.warmup // 2), 2)
        if rank == 0:
            line["two_streams"] = {"value": ms2 / el2, "unit": "depth-maps/s", "ms_per_step": 1e3 * el2 / K, "median_ms_per_step": med2, "steps": K,
                                   "note": f"2 independent forwards of batch {B // 2} per step, one hipGraph + HIP stream each, the model's own (split-f16) layers on both "
                                           "streams; rounds 2-4 had to run such replicas all-float32 (705 depth-maps/s): float32 kernels returned wrong values beside "
                                           "another stream's f16 matrix instructions - one packed-float32 instruction form (tools/probes/pk_fma_opsel_repro.hip), which "
                                           "the library is now assembled without (casmvsnet_pl_amd/build.py); bit equality with the single-stream forward: "
                                           "tests/test_gpu_system.py, tools/gpu_mixed_streams.py"}
    if not args.no_batch1:   # the same configuration with conv0 in the OTHER arithmetics, beside the headline (not instead of it)
        this_mode = args.conv0_mode or "splitf16"
        others = []
        for other in ("f32", "splitbf16", "splitf16"):
            if other == this_mode:
                continue
            conv0_mode[0] = other
            _, _, elo, mso, medo, _ = measure(B, K, max(2, args.warmup // 2), NS)
            others.append({"conv0_mode": other, "value": mso / elo, "unit": "depth-maps/s", "ms_per_step": 1e3 * elo / K, "median_ms_per_step": medo})
        if rank == 0:
            line["conv0_other_modes"] = {"modes": others, "note": "same launch configuration as the headline; 'f32' = every CostRegNet layer on the "
                                                                  "float32 MFMA kernels, 'splitbf16' = only conv0 on the bf16 matrix cores, 'splitf16' = "
                                                                  "conv0, conv2, conv4 on the f16 matrix cores"}
        conv0_mode[0] = None
    if B != 1 and not args.no_batch1:
        m1, in1, el1, mp1, medb1, g1 = measure(1, K, max(2, args.warmup // 2), 1)
        if rank == 0:
            line["batch1"] = {"value": mp1 / el1, "unit": "depth-maps/s", "ms_per_step": 1e3 * el1 / K, "median_ms_per_step": medb1,
                              "steps": K, "launch": "one hipGraph replay per step" if g1 else "kernel by kernel",
                              "note": "eval.py:213-222 processes one reference view per forward (one stream)"}
        if not args.no_events:
            r1 = instrumented_pass(m1, in1, args.config, 1, K, max(1, args.event_every), barrier, dev)
            if rank == 0:
                for k in ("roofline", "roofline_costreg", "roofline_costvol", "roofline_feature", "roofline_homo_warp", "roofline_prob_regress",
                          "roofline_softmax", "stage_ms_per_step"):
                    if k in r1:
                        line["batch1"][k] = r1[k]
        del m1
        if g1:
            n1 = max(2, NS)
            _, _, el1s, mp1s, _, _ = measure(1, K, max(2, args.warmup // 2), n1)
            if rank == 0:
                line["batch1"]["two_streams" if n1 == 2 else "concurrent"] = {
                    "value": mp1s / el1s, "unit": "depth-maps/s", "ms_per_step": 1e3 * el1s / K,
                    "note": f"{n1} single-view forwards in flight, one stream + hipGraph each (the eval.py loop over independent reference views, two at a time)"}
    if world == 1 and args.mode == "replica" and args.config == HEADLINE and not args.no_train_step:
        # f-2 (train.py:99-127) in the driver's line: the batch-1 training step as one hipGraph replay, 20 timed replays (~0.3 s)
        torch.cuda.empty_cache()
        targs = argparse.Namespace(**vars(args))
        targs.batch, targs.batch_given, targs.steps, targs.warmup, targs.no_graph = 1, False, 20, 3, False
        try:
            tl = train_mode(targs, dev, 1, 0, None, barrier)
            line["train_step"] = {"train_step_ms": tl["train_step_ms"], "samples_per_s": tl["value"], "median_ms_per_step": tl.get("median_ms_per_step"),
                                  "steps": 20, "peak_memory_gib": tl["peak_memory_gib"], "config": tl["config"],
                                  "note": "the reference's training step (train.py:99-127: train-mode forward with batch-statistics InPlaceABN, SL1 loss, "
                                          "backward, SGD) on the same 640x512 x 3-view workload, batch 1, one hipGraph replay per step; "
                                          "`python bench.py --mode train` prints it as its own line"}
        except Exception as e:   # the inference line must not be lost to the extra measurement
            line["train_step"] = {"error": f"{type(e).__name__}: {str(e).splitlines()[0][:300]}"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config)
        if world == 1 and args.stock_pytorch:
            line["stock_pytorch_rocm"] = stock_pytorch_rocm(args.config, dev)
        emit(line, args.full_out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
