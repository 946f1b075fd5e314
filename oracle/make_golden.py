"""ORACLE - TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by EXECUTING THE REAL REFERENCE
(/root/reference/models/mvsnet.py, unmodified, + the import shims under oracle/shims) on seeded
synthetic inputs and weights.  Only runnable in the build container; the fixtures it writes are
committed so that the GPU box (which has no /root/reference) can check against them.

    python oracle/make_golden.py            # rewrites every fixture

Inputs and weights are NOT stored: they are regenerated bit-identically from the seed by
casmvsnet_pl_amd.synthetic (same torch build in the container and on the GPU box); each fixture
carries checksums of the regenerated tensors so drift is detected instead of silently compared.
Reference hook points: homo_warp and CostRegNet.forward outputs are captured by wrapping the
reference's own functions at run time (no source edit).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from casmvsnet_pl_amd import ABN, CascadeMVSNet  # noqa: E402  (only for the state-dict key/shape list)
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict, tensor_checksum  # noqa: E402
from oracle.reference_loader import build_reference_model, load_reference  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# name -> config.  Small enough that every per-stage tensor fits in a few MB.
CASES = {
    "e2e_var_v3_32x64": dict(H=32, W=64, V=3, G=1, geometry="dtu", seed=11, wseed=1),
    "e2e_gwc8_v3_32x64": dict(H=32, W=64, V=3, G=8, geometry="dtu", seed=12, wseed=2, prob_gain=(0.3, 3.0, 16.0)),
    "e2e_var_v5_64x96": dict(H=64, W=96, V=5, G=1, geometry="dtu", seed=13, wseed=3, store_volumes=False),
    "e2e_var_v3_random_32x64": dict(H=32, W=64, V=3, G=1, geometry="random", seed=14, wseed=4, store_volumes=False,
                                    prob_gain=(0.8, 8.0, 32.0)),
    "e2e_gwc4_v4_64x64": dict(H=64, W=64, V=4, G=4, geometry="dtu", seed=15, wseed=5, store_volumes=False,
                              interval_ratios=(1.0, 2.5, 5.5), depth_interval=2.0, prob_gain=(0.3, 1.5, 10.0)),
}
N_DEPTHS = (8, 32, 48)


DEFAULT_PROB_GAIN = (0.2, 0.5, 2.0)


def make_state_dict(num_groups, wseed, prob_gain=DEFAULT_PROB_GAIN):
    sd = CascadeMVSNet(n_depths=list(N_DEPTHS), num_groups=num_groups, norm_act=ABN).state_dict()
    return randomize_state_dict(sd, wseed, prob_gain=tuple(prob_gain))


def run_reference(cfg):
    ratios = cfg.get("interval_ratios", (1.0, 2.0, 4.0))
    sd = make_state_dict(cfg["G"], cfg["wseed"], cfg.get("prob_gain", DEFAULT_PROB_GAIN))
    imgs, proj, dmin, dint = make_inputs(1, cfg["V"], cfg["H"], cfg["W"], seed=cfg["seed"], geometry=cfg["geometry"])
    dint = cfg.get("depth_interval", dint)
    model = build_reference_model(N_DEPTHS, ratios, cfg["G"], sd)
    mvsnet, modules, _ = load_reference()
    captured = {"warp": [], "volume": [], "cost": [], "dv": []}
    orig_warp = mvsnet.homo_warp

    def warp_hook(src_feat, proj_mat, depth_values):
        out = orig_warp(src_feat, proj_mat, depth_values)
        captured["warp"].append(out.detach().clone())
        captured["dv"].append(depth_values.detach().clone())
        return out

    hooks = []
    for l in range(3):
        def pre(mod, args):
            captured["volume"].append(args[0].detach().clone())

        def post(mod, args, out):
            captured["cost"].append(out.detach().clone())
        m = getattr(model, f"cost_reg_{l}")
        hooks += [m.register_forward_pre_hook(pre), m.register_forward_hook(post)]
    mvsnet.homo_warp = warp_hook
    try:
        with torch.no_grad():
            res = model(imgs, proj, dmin, dint)
    finally:
        mvsnet.homo_warp = orig_warp
        for h in hooks:
            h.remove()
    V = cfg["V"]
    out = {"meta_n_depths": np.array(N_DEPTHS), "meta_interval_ratios": np.array(ratios, dtype=np.float64),
           "meta_hwvg": np.array([cfg["H"], cfg["W"], V, cfg["G"]]), "meta_seeds": np.array([cfg["seed"], cfg["wseed"]]),
           "meta_geometry": np.array(cfg["geometry"]), "meta_depth": np.array([dmin, dint], dtype=np.float64),
           "meta_prob_gain": np.array(cfg.get("prob_gain", DEFAULT_PROB_GAIN), dtype=np.float64),
           "chk_imgs": np.array(tensor_checksum(imgs)), "chk_proj": np.array(tensor_checksum(proj)),
           "chk_weights": np.array(sum(tensor_checksum(v) for v in sd.values()))}
    for k, v in res.items():
        out[k] = v.numpy()
    for i, l in enumerate((2, 1, 0)):  # call order is coarse -> fine
        out[f"depth_values_{l}"] = captured["dv"][i * (V - 1)].numpy()
        out[f"cost_{l}"] = captured["cost"][i].squeeze(1).numpy()
        vol = captured["volume"][i]
        out[f"chk_volume_{l}"] = np.array(tensor_checksum(vol))
        if cfg.get("store_volumes", True):
            out[f"volume_{l}"] = vol.numpy()
    if cfg.get("store_volumes", True):
        # one un-fused homo_warp output (coarsest level, first source view) for op-level parity
        out["warp_2_v1"] = captured["warp"][0].numpy()
    return out


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    for name, cfg in CASES.items():
        out = run_reference(cfg)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, depth_0 mean {out['depth_0'].mean():.3f}, "
              f"confidence means {[round(float(out[f'confidence_{l}'].mean()), 3) for l in range(3)]}")


if __name__ == "__main__":
    main()
