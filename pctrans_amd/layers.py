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

    def forward(self, x):
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
