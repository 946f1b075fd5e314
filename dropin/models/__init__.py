"""Drop-in for the reference's `models` package: put `<repo>/dropin` (and `<repo>`) on PYTHONPATH and
the reference's `train.py` / `eval.py` imports (`from models.mvsnet import CascadeMVSNet`) resolve
to the MI355X engine.  See INTEGRATION.md."""
