"""TEST INFRASTRUCTURE ONLY - import shim so the *unmodified* reference
(/root/reference/models/modules.py:5 `from inplace_abn import InPlaceABN`) can be
imported in the build container, where the `inplace_abn` CUDA extension is not
installable.  Never imported by the product package.

Semantics restated from the public mapillary/inplace_abn `ABN` module (unpinned
in the reference, README.md:28): BatchNorm(eps=1e-5, momentum=0.1, affine) followed
by leaky_relu(activation_param=0.01).  Eval mode uses the running statistics.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ABN(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 activation="leaky_relu", activation_param=0.01):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.affine = affine
        self.activation = activation
        self.activation_param = activation_param
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        x = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                         self.training, self.momentum, self.eps)
        if self.activation == "leaky_relu":
            return F.leaky_relu(x, negative_slope=self.activation_param)
        if self.activation == "relu":
            return F.relu(x)
        if self.activation == "identity":
            return x
        raise RuntimeError(f"unknown activation {self.activation}")


InPlaceABN = ABN
InPlaceABNSync = ABN
