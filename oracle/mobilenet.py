"""Oracle: MobileNetClassifier ("mobilenet", BASELINE configs[4]).  Test infrastructure only.

Restates ``howl/model/cnn.py:15-29``: ``downsample`` = Conv2d(1, 3, 3, padding=(1, 3)) + BatchNorm2d(3) + ReLU +
MaxPool2d((1, 2)), then torchvision's ``mobilenet_v2`` with a fresh ``num_labels`` classifier, applied to ``x[:, :1]``.

PARITY UNPINNED for the MobileNetV2 body: torchvision (``requirements.txt:17``, ``torchvision>=0.6.0``) is a third-party
dependency that is absent from /root/reference and from this image, and the reference loads ImageNet weights over the
network (``mobilenet_v2(pretrained=True)``).  The body below is restated from the published architecture
(Sandler et al. 2018, table 2; torchvision ``mobilenetv2.py``: width 1.0, round_nearest 8, ConvBNReLU = Conv(bias=False) +
BatchNorm2d + ReLU6, classifier = Dropout(0.2) + Linear(1280, num_classes)) with torch CPU ops; only the reference's own
call site and the state-dict key names anchor it.  Weights are random-init / closed-form only.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
DROPOUT_P = 0.2
LAST_CHANNEL = 1280
# t (expansion), c (output channels), n (repeats), s (stride of the first repeat)
INVERTED_RESIDUAL_SETTING = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                             [6, 320, 1, 1]]


def layer_table() -> List[dict]:
    """Conv+BN(+act) layers in execution order.  ``key`` is the conv's state-dict prefix, ``bn`` the BatchNorm's.
    ``res`` marks the projection layers whose output adds the block input."""
    layers = [dict(key="downsample.0", bn="downsample.1", kind="dense", cin=1, cout=3, stride=1, pad=(1, 3), act="relu",
                   bias=True, pool=True, res=False, block_first=False),
              dict(key="model.features.0.0", bn="model.features.0.1", kind="dense", cin=3, cout=32, stride=2, pad=(1, 1),
                   act="relu6", bias=False, pool=False, res=False, block_first=False)]
    inp = 32
    idx = 1
    for t, c, n, s in INVERTED_RESIDUAL_SETTING:
        for i in range(n):
            stride = s if i == 0 else 1
            hidden = inp * t
            pre = f"model.features.{idx}.conv"
            j = 0
            first = True
            if t != 1:
                layers.append(dict(key=f"{pre}.0.0", bn=f"{pre}.0.1", kind="pw", cin=inp, cout=hidden, stride=1, pad=(0, 0),
                                   act="relu6", bias=False, pool=False, res=False, block_first=True))
                j = 1
                first = False
            layers.append(dict(key=f"{pre}.{j}.0", bn=f"{pre}.{j}.1", kind="dw", cin=hidden, cout=hidden, stride=stride,
                               pad=(1, 1), act="relu6", bias=False, pool=False, res=False, block_first=first))
            layers.append(dict(key=f"{pre}.{j + 1}", bn=f"{pre}.{j + 2}", kind="pw", cin=hidden, cout=c, stride=1, pad=(0, 0),
                               act="none", bias=False, pool=False, res=(stride == 1 and inp == c), block_first=False))
            inp = c
            idx += 1
    layers.append(dict(key="model.features.18.0", bn="model.features.18.1", kind="pw", cin=inp, cout=LAST_CHANNEL, stride=1,
                       pad=(0, 0), act="relu6", bias=False, pool=False, res=False, block_first=False))
    return layers


def conv_shape(l):
    if l["kind"] == "dense":
        return (l["cout"], l["cin"], 3, 3)
    if l["kind"] == "dw":
        return (l["cout"], 1, 3, 3)
    return (l["cout"], l["cin"], 1, 1)


def mobilenet_init(num_labels: int, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded parameters with torchvision's initial distributions (kaiming-normal fan-out convolutions, N(0, 0.01)
    classifier) but non-trivial BatchNorm affine terms and conv bias, so that every gradient path is exercised; fresh BN
    buffers.  (Closed-form sinusoid weights as used for res8 are rank-2 as matrices: the 1x1 convolutions would feed
    BatchNorm degenerate channels.)"""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for l in layer_table():
        shape = conv_shape(l)
        fan_out = shape[0] * shape[2] * shape[3] // (l["cin"] if l["kind"] == "dw" else 1)
        sd[l["key"] + ".weight"] = torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_out)
        if l["bias"]:
            sd[l["key"] + ".bias"] = torch.randn(l["cout"], generator=gen) * 0.1
        sd[l["bn"] + ".weight"] = 1.0 + 0.2 * torch.randn(l["cout"], generator=gen)
        sd[l["bn"] + ".bias"] = 0.1 * torch.randn(l["cout"], generator=gen)
        sd[l["bn"] + ".running_mean"] = torch.zeros(l["cout"])
        sd[l["bn"] + ".running_var"] = torch.ones(l["cout"])
        sd[l["bn"] + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    sd["model.classifier.1.weight"] = torch.randn((num_labels, LAST_CHANNEL), generator=gen) * 0.05
    sd["model.classifier.1.bias"] = torch.randn(num_labels, generator=gen) * 0.1
    return sd


def mobilenet_param_names() -> List[str]:
    names = []
    for l in layer_table():
        names.append(l["key"] + ".weight")
        if l["bias"]:
            names.append(l["key"] + ".bias")
        names += [l["bn"] + ".weight", l["bn"] + ".bias"]
    return names + ["model.classifier.1.weight", "model.classifier.1.bias"]


def mobilenet_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, training: bool,
                      keep_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``MobileNetClassifier.forward`` (``cnn.py:26-29``).  x: (B, C>=1, M, T).  ``keep_mask`` (B, 1280) of 0/1 replaces
    the Dropout(0.2) draw of training mode (kept activations are scaled by 1/(1-p)); None = no dropout."""
    x = x[:, :1]
    block_in = None
    for l in layer_table():
        if l["block_first"]:
            block_in = x
        w = sd[l["key"] + ".weight"]
        b = sd.get(l["key"] + ".bias") if l["bias"] else None
        groups = l["cin"] if l["kind"] == "dw" else 1
        z = F.conv2d(x, w, b, stride=l["stride"], padding=l["pad"], groups=groups)
        rm, rv = sd[l["bn"] + ".running_mean"], sd[l["bn"] + ".running_var"]
        y = F.batch_norm(z, rm, rv, sd[l["bn"] + ".weight"], sd[l["bn"] + ".bias"], training, BN_MOMENTUM, BN_EPS)
        if training:
            sd[l["bn"] + ".num_batches_tracked"] += 1
        if l["act"] == "relu6":
            y = F.relu6(y)
        elif l["act"] == "relu":
            y = F.relu(y)
        if l["res"]:
            y = y + block_in
        if l["pool"]:
            y = F.max_pool2d(y, (1, 2))
        x = y
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    if keep_mask is not None:
        x = x * keep_mask / (1.0 - DROPOUT_P)
    return F.linear(x, sd["model.classifier.1.weight"], sd["model.classifier.1.bias"])
