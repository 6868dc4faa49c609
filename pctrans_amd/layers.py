"""Small stand-ins for the detectron2 / fvcore helpers the reference head uses (neither is vendored nor installed):
`ShapeSpec`, `Conv2d` (conv + optional norm + optional activation, keys `weight`, `bias`, `norm.*`), `get_norm`,
`c2_xavier_fill`.  Used at msdeformattn.py:13-15,265-288 and mask2former_transformer_decoder.py:11,19,387-388 of the
reference; parameter names are kept so state-dicts interchange."""
from collections import namedtuple

import torch
from torch import nn
from torch.nn import functional as F


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm with fixed statistics and affine (buffers, not parameters) -- y = x * scale + shift."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.to(x.dtype).view(1, -1, 1, 1) + shift.to(x.dtype).view(1, -1, 1, 1)


class LayerNorm2d(nn.Module):
    """LayerNorm over the channel dim of NCHW tensors (detectron2's "LN")."""

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


def get_norm(norm, out_channels):
    """"" / None -> no norm; "BN", "SyncBN", "FrozenBN", "GN" (32 groups), "LN"; or a callable(channels)."""
    if norm is None:
        return None
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        norm = {
            "BN": nn.BatchNorm2d,
            "SyncBN": nn.SyncBatchNorm,       # RCCL all_gather of batch stats under DDP; running stats in eval
            "FrozenBN": FrozenBatchNorm2d,
            "GN": lambda channels: nn.GroupNorm(32, channels),
            "LN": LayerNorm2d,
        }[norm]
    return norm(out_channels)


class Conv2d(nn.Conv2d):
    """nn.Conv2d with optional `norm` sub-module and `activation` callable applied after it."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def _pointwise(self, x):
        return (x.is_cuda and x.dim() == 4 and self.kernel_size == (1, 1) and self.stride == (1, 1)
                and self.padding == (0, 0) and self.dilation == (1, 1) and self.groups == 1
                and self.padding_mode == "zeros" and x.is_contiguous())

    def forward(self, x):
        if self._pointwise(x):
            # a 1x1 convolution on NCHW data is W[Cout,Cin] @ x[n][Cin,HW]: one strided-batched GEMM, no layout
            # transposes and no dependence on MIOpen's per-process solver search (autocast treats it like conv2d)
            n, c, h, w = x.shape
            wm = self.weight.view(1, self.out_channels, c).expand(n, -1, -1)
            x3 = x.view(n, c, h * w)
            if self.bias is not None:
                y = torch.baddbmm(self.bias.view(1, -1, 1), wm, x3)
            else:
                y = torch.bmm(wm, x3)
            x = y.view(n, self.out_channels, h, w)
        else:
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def c2_xavier_fill(module):
    """Caffe2 XavierFill == kaiming_uniform_(a=1); zero bias."""
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class CachedLinear(nn.Linear):
    """nn.Linear with the same parameters / state-dict keys; under CUDA autocast in forward-only use it keeps
    low-precision copies of weight and bias instead of re-casting them on every forward (autocast's own cache lives only
    for one autocast region: ~330 cast launches per decoder forward), and can apply ReLU in the GEMM epilogue.  The copies
    are keyed on the parameters' version counters and storage, so optimizer steps, `load_state_dict` and `.to()` refresh
    them.  Writes through `.data` (e.g. an EMA swap `p.data.copy_(...)`) do NOT bump the version counter: call
    `invalidate()` after such a write.  Anything else (fp32, CPU, autograd) is plain `nn.Linear`."""

    _lp_cache = None

    def invalidate(self):
        """Drop the cached low-precision copies (after a write the version counters cannot see)."""
        self._lp_cache = None

    def _low_precision(self, dtype):
        w, b = self.weight, self.bias
        key = (dtype, w._version, w.data_ptr(), -1 if b is None else b._version, 0 if b is None else b.data_ptr())
        c = self._lp_cache
        if c is None or c[0] != key:
            with torch.no_grad():
                c = (key, w.detach().to(dtype), None if b is None else b.detach().to(dtype))
            self._lp_cache = c
        return c[1], c[2]

    def forward(self, x, relu=False):
        if (x.is_cuda and torch.is_autocast_enabled("cuda")
                and not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad))):
            dtype = torch.get_autocast_dtype("cuda")
            w, b = self._low_precision(dtype)
            x = x if x.dtype == dtype else x.to(dtype)
            if relu and b is not None:
                x2 = x.reshape(-1, x.shape[-1])
                return torch._addmm_activation(b, x2, w.t(), use_gelu=False).view(*x.shape[:-1], w.shape[0])
            y = F.linear(x, w, b)
            return F.relu(y) if relu else y
        y = super().forward(x)
        return F.relu(y) if relu else y
