"""`inplace_abn` compatibility module (pure PyTorch parameter container).

The reference passes `inplace_abn.InPlaceABN` (train.py:10,41) or `inplace_abn.ABN`
(eval.py:13,201) as `norm_act`; that CUDA extension cannot be built for ROCm here.  These classes
keep the constructor, parameter/buffer names (`weight`, `bias`, `running_mean`, `running_var`) and
attributes (`eps`, `activation`, `activation_param`) the checkpoints and the weight folding in
`casmvsnet_pl_amd.mvsnet` rely on.  On the MI355X inference path the CostRegNet ABNs are never
executed as modules: they are folded into the MFMA conv epilogue (scale, shift, leaky slope).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ABN(nn.Module):
    """BatchNorm (eps 1e-5) followed by an activation (leaky_relu 0.01 by default)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 activation="leaky_relu", activation_param=0.01):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.affine = affine
        self.activation = activation
        self.activation_param = activation_param
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def leaky_slope(self):
        """Negative-side slope of the activation as the conv epilogue applies it."""
        if self.activation == "leaky_relu":
            return float(self.activation_param)
        if self.activation == "relu":
            return 0.0
        if self.activation == "identity":
            return 1.0
        raise RuntimeError(f"unsupported ABN activation {self.activation!r}")

    def folded_scale_shift(self):
        """Eval-mode BN as y = x * scale + shift (float64 math, float32 result)."""
        var = self.running_var.detach().double()
        mean = self.running_mean.detach().double()
        gamma = self.weight.detach().double() if self.affine else torch.ones_like(var)
        beta = self.bias.detach().double() if self.affine else torch.zeros_like(var)
        scale = gamma / torch.sqrt(var + self.eps)
        shift = beta - mean * scale
        return scale.float().cpu(), shift.float().cpu()

    def forward(self, x):
        x = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                         self.training, self.momentum, self.eps)
        if self.activation == "leaky_relu":
            return F.leaky_relu(x, negative_slope=self.activation_param)
        if self.activation == "relu":
            return F.relu(x)
        if self.activation == "identity":
            return x
        raise RuntimeError(f"unsupported ABN activation {self.activation!r}")

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, activation={self.activation}[{self.activation_param}]"


class InPlaceABN(ABN):
    """The in-place memory trick of mapillary's CUDA extension is irrelevant here, its ARITHMETIC is not: to keep the
    activation invertible the extension normalises with gamma = |weight| + eps instead of `weight` (as recalled from
    the upstream kernels and the In-Place ABN paper - the extension cannot be built or inspected offline).  A negative
    or tiny BN weight therefore gives a different layer under `norm_act=InPlaceABN` (train.py:41, Lightning
    validation) than under `norm_act=ABN` (eval.py:201, what the oracle and the golden fixtures run): both classes
    reproduce their own upstream semantics here, in the module forward and in the folded conv epilogue."""

    def _gamma(self):
        return self.weight.abs() + self.eps

    def folded_scale_shift(self):
        if not self.affine:
            return super().folded_scale_shift()
        var = self.running_var.detach().double()
        mean = self.running_mean.detach().double()
        gamma = self.weight.detach().double().abs() + self.eps
        scale = gamma / torch.sqrt(var + self.eps)
        shift = self.bias.detach().double() - mean * scale
        return scale.float().cpu(), shift.float().cpu()

    def forward(self, x):
        if not self.affine:
            return super().forward(x)
        x = F.batch_norm(x, self.running_mean, self.running_var, self._gamma(), self.bias,
                         self.training, self.momentum, self.eps)
        if self.activation == "leaky_relu":
            return F.leaky_relu(x, negative_slope=self.activation_param)
        if self.activation == "relu":
            return F.relu(x)
        if self.activation == "identity":
            return x
        raise RuntimeError(f"unsupported ABN activation {self.activation!r}")


InPlaceABNSync = InPlaceABN
