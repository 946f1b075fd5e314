"""CPU checks of bench.py's bookkeeping: the algorithmic bytes / FLOPs it divides by are the ones SURVEY 8(d)
states, and it refuses to run without a GPU instead of falling back."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_matches_survey_8d(bench):
    H, W, V, G, n_depths = bench.CONFIGS["dtu_640x512_v3_var"][:5]
    work = bench.algorithmic_work(H, W, V, G, n_depths)
    # fused cost-volume build 137.6 / 194.0 / 125.8 MB at levels 2 / 1 / 0 (work is keyed by level)
    assert [round(work[l]["costvol_bytes"] / 1e6, 1) for l in (2, 1, 0)] == [137.6, 194.0, 125.8]
    assert [round(work[l]["softmax_bytes"] / 1e6, 1) for l in (2, 1, 0)] == [8.0, 21.6, 23.6]
    # CostRegNet 19.96 / 35.11 / 26.05 GFLOP, conv0 alone 13.59 / 18.12 / 9.06
    assert [round(work[l]["costreg_flops"] / 1e9, 2) for l in (2, 1, 0)] == [19.96, 35.11, 26.05]
    assert [round(work[l]["conv0_flops"] / 1e9, 2) for l in (2, 1, 0)] == [13.59, 18.12, 9.06]
    gwc = bench.algorithmic_work(*bench.CONFIGS["dtu_640x512_v3_gwc8"][:5])
    assert [round(gwc[l]["costvol_bytes"] / 1e6, 1) for l in (2, 1, 0)] == [43.3, 110.1, 125.8]
    # the un-fused homo_warp op: 132.4 / 183.5 / 104.9 MB (one source view)
    assert [round(work[l]["homo_warp_bytes"] / 1e6, 1) for l in (2, 1, 0)] == [132.4, 183.5, 104.9]
    # FeatureNet: 16.9 GFLOP for the 3 views
    assert round(3 * bench.feature_flops(H, W) / 1e9, 1) == 16.9


def test_pmc_traffic_lookup(bench):
    val, note, src = bench.pmc_traffic("conv16db_kernel<2, 4, 4, 4, 4, 32", 2)
    assert val is None or 3.8e8 < val < 2e9, note   # >= the algorithmic 384 MB of the same launches
    assert val is None or src["file"].startswith("profiles/")
    assert bench.pmc_traffic("conv16db_kernel<2, 4, 4, 4, 4, 32", 1)[0] is None


def _rank_aggregate(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # rank r took (1 + r) seconds for 40 * (r + 1) depth maps
    rep = mod.aggregate(1.0 + rank, 40 * (rank + 1), dist)                      # replica: MAX time, SUM maps
    shard = mod.aggregate(1.0 + rank, 40, dist, sum_maps=False)                 # view-sharded: the same maps on every rank
    line = None
    if rank == 0:
        line = mod.base_line("depth-maps/sec at 640x512, 3 views, n_depths=[8,32,48]", "depth-maps/s", rep[1] / rep[0], world, 20, 5,
                             rep[0], "weak", {"workload": "dtu_640x512_v3_var"}, 1.23)
    q.put((rank, rep, shard, line))
    dist.destroy_process_group()


def test_rank_aggregation_and_json_contract_at_world_size_2():
    """bench.py's cross-rank bookkeeping on the gloo backend (the driver's N > 1 runs use nccl = RCCL): wall time = MAX over
    ranks, depth maps = SUM over ranks (replica) or the local count (view-sharded); rank 0's line carries the driver's keys."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_rank_aggregate, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, rep, shard, line in got:
        assert rep == (2.0, 120) and shard == (2.0, 40)
    line = got[0][3]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config"):
        assert key in line
    assert line["value"] == 60.0 and line["n_gpus"] == 2 and line["ms_per_step"] == 100.0 and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["dtype"] == "f32" and line["config"]["workload"] == "dtu_640x512_v3_var"
    assert bench_single_rank_aggregate()


def bench_single_rank_aggregate():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.aggregate(0.5, 7) == (0.5, 7)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_gpus_flag_builds_the_contract_launch_line_and_refuses_missing_gpus(bench):
    """`python bench.py --gpus N` without WORLD_SIZE starts the N ranks itself under torch.distributed.run (one process per GPU, 127.0.0.1
    rendezvous) - the driver's own launch line; on a node with fewer GPUs it says so instead of silently running world size 1."""
    cmd = bench.torchrun_command(4, ["--gpus", "4", "--steps", "3"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "this node shows 0 GPU(s)" in (r.stderr + r.stdout)


def test_dominant_kernel_roofline_never_exceeds_its_roof(bench):
    """The arithmetic behind `roofline` for the split-f16 conv0: algorithmic bytes / time against 8 TB/s, the f16 FLOPs the matrix cores execute
    against the dense f16 peak.  With round 3's driver-timed launch times (profiles/r03_kernel_stats.csv: 1002.6 / 646.2 / 486.3 us at levels 1 / 2 / 0,
    batch 8) both fractions are ~0.25 - the float32-peak ratio that read 0.99 is reported as `fp32_equivalent`, not as a fraction."""
    H, W, V, G, n_depths = bench.CONFIGS["dtu_640x512_v3_var"][:5]
    B = 8
    work = bench.algorithmic_work(H, W, V, G, n_depths, B)
    t = {1: 1002.6e-6, 2: 646.2e-6, 0: 486.3e-6}
    alg_bytes = sum(4 * B * (8 * 2 ** l + 8) * n_depths[l] * (H >> l) * (W >> l) for l in range(3))
    flops = sum(work[l]["conv0_flops"] for l in range(3))
    hbm_frac = alg_bytes / sum(t.values()) / 1e9 / bench.HBM_PEAK_GBS
    f16_frac = 4.0 * flops / sum(t.values()) / 1e12 / bench.MFMA_F16_PEAK_TFLOPS
    assert 0.2 < hbm_frac < 0.3 and 0.2 < f16_frac < 0.3
    assert flops / sum(t.values()) / 1e12 / bench.MFMA_F32_PEAK_TFLOPS > 0.9   # the old headline ratio: why it is no longer called `frac`


def _worst_case_full_line():
    """Round 5's full bench object (the 21.5 kB line the driver could not parse), with every string doubled and extra objects added: a line that only grows."""
    import json
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))

    def grow(o):
        if isinstance(o, dict):
            return {k: grow(v) for k, v in o.items()}
        if isinstance(o, str) and len(o) > 24:   # prose and kernel descriptions, not the enumerations ("hbm", "GB/s", "port")
            return o + " " + o
        return o
    full = grow(full)
    full["single_stream"] = dict(full["two_streams"])
    full["stock_pytorch_rocm"] = {"value": 12.3, "unit": "depth-maps/s", "sample": "x" * 500}
    full["train_step"]["error"] = "RuntimeError: " + "y" * 1000
    for i in range(50):
        full[f"future_object_{i}"] = {"note": "z" * 300}
    return full


def test_stdout_line_stays_under_4_kb_and_keeps_what_the_driver_reads(bench):
    """BENCH_r05.json.parsed was null: the stdout line had grown to 21.5 kB.  The line is now built by compact_line(): at most 4 kB whatever the full
    object holds, with the contract keys, `roofline` (frac, traffic), `cpu_baseline` and `config.workload`; the full object goes to a side file."""
    import json
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    for obj in (full, _worst_case_full_line()):
        line = bench.compact_line(obj)
        text = json.dumps(line)
        assert len(text) < 4096, len(text)
        assert "\n" not in text
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert key in line, key
        assert line["config"]["workload"] == "dtu_640x512_v3_var" and "model" not in line["config"]
        assert len(line["dtype"]) <= 120 and line["dtype"].startswith("f32")
        rf = line["roofline"]
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and len(rf["kernel"]) <= 80
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0.0 < rf["frac"] < 1.0
        assert rf["traffic"] is None or rf["traffic"] >= 0.95 * rf["algorithmic_bytes_per_launch"]
        cb = line["cpu_baseline"]
        assert cb["value"] > 0 and cb["unit"] == "depth-maps/s" and cb["cores"] >= 1 and cb["kind"] in ("port", "reference")
        assert abs(line["value"] - full["value"]) / full["value"] < 1e-4          # 5 significant digits
        assert abs(line["ms_per_step"] * line["value"] / 1e3 - 8.0) < 1e-2        # value = batch / time: the driver's consistency check
        for key in ("roofline_homo_warp_frac", "roofline_homo_warp_frac_hot", "roofline_costvol_frac", "roofline_costreg_frac_executed",
                    "roofline_costreg_frac_all_float32", "batch1_value", "batch1_two_streams_value", "all_float32_value", "train_step_ms"):
            assert isinstance(line[key], float), key
        assert all(not isinstance(v, (dict, list)) for k, v in line.items() if k not in ("config", "roofline", "cpu_baseline"))


def test_emit_prints_one_parsable_line_and_writes_the_full_object(bench, tmp_path, capsys):
    import json
    full = _worst_case_full_line()
    path = tmp_path / "sub" / "bench_full.json"
    bench.emit(full, str(path))
    out, err = capsys.readouterr()
    lines = [l for l in out.split("\n") if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["roofline"]["frac"] > 0
    assert json.load(open(path))["stage_ms_per_step"]            # the stage table lives in the side file
    assert err.startswith("bench_full: ")
    bench.emit(full, "/proc/definitely/not/writable.json")        # a read-only tree does not lose the line
    out, err = capsys.readouterr()
    assert json.loads(out.strip())["value"] > 0 and "warning: could not write" in err
