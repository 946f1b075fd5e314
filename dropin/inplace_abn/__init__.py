"""Drop-in for `inplace_abn` (train.py:10 `InPlaceABN`, eval.py:13 `ABN`) - see
casmvsnet_pl_amd/inplace_abn.py."""
from casmvsnet_pl_amd.inplace_abn import ABN, InPlaceABN, InPlaceABNSync  # noqa: F401
