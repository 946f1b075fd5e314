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
