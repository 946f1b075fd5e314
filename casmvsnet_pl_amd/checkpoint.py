"""Checkpoints the way the reference's scripts use them (utils/__init__.py:51-80; eval.py:202-205, train.py:47-52):
a plain model state dict, or a PyTorch-Lightning checkpoint whose `state_dict` carries the network under the `model.`
prefix.  The 206 state-dict keys are the reference's, so the released checkpoints load into this engine unchanged."""
import torch


def extract_model_state_dict(ckpt_path, prefixes_to_ignore=(), trust_checkpoint=False):
    """utils/__init__.py:51-74: -> {key: tensor} of the network (Lightning's `model.` prefix stripped, keys that start
    with one of `prefixes_to_ignore` dropped).  The file is read with `weights_only=True` (tensors and primitive
    containers only - what the released checkpoints hold); `trust_checkpoint=True` allows arbitrary pickled objects
    (a Lightning checkpoint with custom hyper-parameter objects) and must only be used on files you produced."""
    checkpoint = torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=not trust_checkpoint)
    source = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    lightning = "state_dict" in checkpoint
    out = {}
    for k, v in source.items():
        if lightning:
            if not k.startswith("model."):
                continue
            k = k[6:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k] = v
    return out


def load_ckpt(model, ckpt_path, prefixes_to_ignore=(), trust_checkpoint=False):
    """utils/__init__.py:76-80: update the model's own state dict with the checkpoint's entries, then load it (strict)."""
    state = model.state_dict()
    state.update(extract_model_state_dict(ckpt_path, prefixes_to_ignore, trust_checkpoint))
    model.load_state_dict(state)
    return model


def save_ckpt(model, ckpt_path, lightning_layout=True, **extra):
    """Writes what load_ckpt reads: {"state_dict": {"model.<key>": tensor}, **extra} (the layout of the reference's
    Lightning checkpoints, e.g. extra = dict(epoch=..., optimizer_states=[...])) or the plain state dict."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    obj = dict(state_dict={"model." + k: v for k, v in sd.items()}, **extra) if lightning_layout else sd
    torch.save(obj, ckpt_path)
