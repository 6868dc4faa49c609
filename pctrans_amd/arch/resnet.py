"""Plain ResNet (18 / 50) that emits {"res2".."res5"} -- the backbone contract MaskFormer needs
(the reference calls detectron2.modeling.build_backbone -> build_resnet_backbone, arch/maskformer.py:9,74; detectron2
is not vendored).  Convolutions stay on MIOpen; normalisation defaults to FrozenBN like the reference config
(config/maskfoermer_config.py:69).  Out of the hot-path scope: only here so the meta-arch runs end to end."""
from torch import nn
from torch.nn import functional as F

from ..layers import ShapeSpec, get_norm


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, mid, stride, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, mid, 3, stride, 1, bias=False)
        self.n1 = get_norm(norm, mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, 1, 1, bias=False)
        self.n2 = get_norm(norm, mid)
        self.short = None
        if stride != 1 or cin != mid:
            self.short = nn.Sequential(nn.Conv2d(cin, mid, 1, stride, bias=False), get_norm(norm, mid))

    def forward(self, x):
        y = F.relu(self.n1(self.conv1(x)))
        y = self.n2(self.conv2(y))
        return F.relu(y + (x if self.short is None else self.short(x)))


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, mid, stride, norm):
        super().__init__()
        cout = mid * 4
        self.conv1 = nn.Conv2d(cin, mid, 1, 1, bias=False)
        self.n1 = get_norm(norm, mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False)      # stride in the 3x3 (STRIDE_IN_1X1: False)
        self.n2 = get_norm(norm, mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, 1, bias=False)
        self.n3 = get_norm(norm, cout)
        self.short = None
        if stride != 1 or cin != cout:
            self.short = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), get_norm(norm, cout))

    def forward(self, x):
        y = F.relu(self.n1(self.conv1(x)))
        y = F.relu(self.n2(self.conv2(y)))
        y = self.n3(self.conv3(y))
        return F.relu(y + (x if self.short is None else self.short(x)))


class ResNet(nn.Module):
    def __init__(self, depth=50, in_channels=3, norm="FrozenBN"):
        super().__init__()
        block, layers = {18: (_Basic, [2, 2, 2, 2]), 34: (_Basic, [3, 4, 6, 3]), 50: (_Bottleneck, [3, 4, 6, 3]),
                         101: (_Bottleneck, [3, 4, 23, 3])}[depth]
        self.stem = nn.Sequential(nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False), get_norm(norm, 64))
        cin, self._shapes, stages = 64, {}, []
        for i, (mid, n) in enumerate(zip([64, 128, 256, 512], layers)):
            blocks = []
            for j in range(n):
                blocks.append(block(cin, mid, (1 if i == 0 else 2) if j == 0 else 1, norm))
                cin = mid * block.expansion
            stages.append(nn.Sequential(*blocks))
            self._shapes["res%d" % (i + 2)] = ShapeSpec(channels=cin, stride=4 * 2 ** i)
        self.res2, self.res3, self.res4, self.res5 = stages

    def output_shape(self):
        return dict(self._shapes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.stem(x)), 3, 2, 1)
        out = {}
        for name in ("res2", "res3", "res4", "res5"):
            x = getattr(self, name)(x)
            out[name] = x
        return out
