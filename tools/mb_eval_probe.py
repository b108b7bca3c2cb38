"""Diagnostic: MobileNet forward in training mode (BatchNorm statistics reduced inside the convolution launches) and in eval mode
(no reductions) on the same batch -- run under rocprofv3 --kernel-trace to compare the same kernels with and without their
arrival / fold tails.   python tools/mb_eval_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from howl_amd.model.cnn import MobileNetClassifier  # noqa: E402

dev = torch.device("cuda:0")
model = MobileNetClassifier(12).to(dev)
x = torch.randn(512, 3, 40, 101, device=dev)
with torch.no_grad():
    for mode in ("train", "eval", "train", "eval"):
        model.train(mode == "train")
        for _ in range(4):
            model._launch_forward(x)
        torch.cuda.synchronize()
print("ok")
