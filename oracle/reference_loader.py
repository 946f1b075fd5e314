"""TEST INFRASTRUCTURE ONLY.

Imports the *unmodified* reference `models/mvsnet.py` from /root/reference with the two
import shims under oracle/shims (inplace_abn, kornia).  Only usable in the build container
(/root/reference does not exist on the GPU box); used by oracle/make_golden.py and by the
CPU tests that pin the restatement (oracle/cpu_restatement.py) to the real reference.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("CASMVS_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "mvsnet.py"))


def load_reference():
    """Return (mvsnet_module, modules_module, ABN) of the real reference."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    if "_casmvs_ref_mvsnet" in sys.modules:  # import once: callers patch/hook the same module object
        return sys.modules["_casmvs_ref_mvsnet"], sys.modules["_casmvs_ref_modules"], sys.modules["_casmvs_ref_abn"].ABN
    saved_path = list(sys.path)
    saved_models = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved_models:
        del sys.modules[k]
    try:
        sys.path.insert(0, REFERENCE_ROOT)
        sys.path.insert(0, _SHIMS)
        for k in ("inplace_abn", "kornia", "kornia.utils"):
            sys.modules.pop(k, None)
        mvsnet = importlib.import_module("models.mvsnet")
        modules = importlib.import_module("models.modules")
        abn_mod = importlib.import_module("inplace_abn")
        abn = abn_mod.ABN
        sys.modules["_casmvs_ref_abn"] = abn_mod
        # keep private handles, then drop the generic names so the product's own
        # `models` / `inplace_abn` drop-in packages can still be imported afterwards.
        sys.modules["_casmvs_ref_mvsnet"] = mvsnet
        sys.modules["_casmvs_ref_modules"] = modules
    finally:
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        for k in ("inplace_abn", "kornia", "kornia.utils"):
            sys.modules.pop(k, None)
        sys.modules.update(saved_models)
        sys.path[:] = saved_path
    return mvsnet, modules, abn


def build_reference_model(n_depths, interval_ratios, num_groups, state_dict=None):
    """Reference CascadeMVSNet in the eval configuration of eval.py:198-205.

    For G=1 the top-level flag is flipped back to training=True (children stay in eval) so that
    mvsnet.py:152-153 (out-of-place accumulate) runs instead of :155, which raises on the
    stride-0 expanded `ref_volume` under torch>=2.  Arithmetic is identical.
    """
    mvsnet, _, abn = load_reference()
    model = mvsnet.CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(interval_ratios),
                                 num_groups=num_groups, norm_act=abn)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    model.eval()
    if num_groups == 1:
        model.training = True
    return model


_GENERIC = ("models", "datasets", "utils", "inplace_abn", "kornia", "cv2", "numba", "plyfile", "torchvision")


def _import_from_reference(names, extra_files=()):
    """Import top-level modules `names` (and single files) of the reference with /root/reference and the shims in front of
    sys.path, keep them under private names, then restore sys.path / sys.modules so that same-named packages of the
    environment (HuggingFace `datasets`, the product's drop-in `models`) stay importable afterwards."""
    import importlib.util
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    saved_path = list(sys.path)
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _GENERIC}
    for k in saved:
        del sys.modules[k]
    out = {}
    try:
        sys.path.insert(0, REFERENCE_ROOT)
        sys.path.insert(0, _SHIMS)
        for n in names:
            out[n] = importlib.import_module(n)
        for alias, rel in extra_files:
            spec = importlib.util.spec_from_file_location(alias, os.path.join(REFERENCE_ROOT, rel))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            out[alias] = mod
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in _GENERIC]:
            del sys.modules[k]
        sys.modules.update(saved)
        sys.path[:] = saved_path
    return out


def load_reference_eval():
    """The unmodified /root/reference/eval.py as a module (its __main__ block does not run): `xy_ref2src`, `xy_src2ref`,
    `check_geo_consistency` (eval.py:113-182) are the reference's own numpy code; behind them `numba.jit` is the identity and
    `cv2.remap` / `cv2.resize` are oracle/fusion_restatement.py (see oracle/shims/cv2)."""
    if "_casmvs_ref_eval" not in sys.modules:
        sys.modules["_casmvs_ref_eval"] = _import_from_reference((), [("_casmvs_ref_eval", "eval.py")])["_casmvs_ref_eval"]
    return sys.modules["_casmvs_ref_eval"]


def load_reference_datasets():
    """The unmodified /root/reference/datasets package (DTUDataset, BlendedMVSDataset, TanksDataset) behind the cv2 /
    torchvision shims (PIL is installed and does the decoding / bilinear resize, as in the reference)."""
    if "_casmvs_ref_datasets" not in sys.modules:
        sys.modules["_casmvs_ref_datasets"] = _import_from_reference(("datasets",))["datasets"]
    return sys.modules["_casmvs_ref_datasets"]
