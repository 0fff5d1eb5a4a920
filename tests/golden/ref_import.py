"""Import harness for the upstream reference (/root/reference) -- development container only.

The reference is Python; it needs a handful of third-party modules that are not in this
image (cv2, yacs, termcolor, torchvision, MinkowskiEngine).  SURVEY.md 8(c) lists the stub
set; this file installs exactly that set into ``sys.modules`` and then imports the
reference from where it lies.  Nothing from the reference is copied.

Only ``tests/golden/make_golden.py`` (the fixture generator) and the optional
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent, i.e. on
the GPU box) use this module.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MTE_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "packnet_code", "packnet_sfm"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def install_stubs():
    import numpy as np
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    if "cv2" not in sys.modules:
        _stub("cv2")
    if "termcolor" not in sys.modules:
        _stub("termcolor", colored=lambda s, *a, **k: s)
    if "yacs" not in sys.modules:
        y = _stub("yacs")
        y.config = _stub("yacs.config", CfgNode=dict)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
    if "MinkowskiEngine" not in sys.modules:
        class _NoOp(nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

            def forward(self, x):
                return x
        _stub("MinkowskiEngine", MinkowskiConvolution=_NoOp, MinkowskiBatchNorm=_NoOp,
              MinkowskiReLU=_NoOp, MinkowskiMaxPooling=_NoOp, MinkowskiSigmoid=_NoOp,
              SparseTensor=object)
    import matplotlib.cm as cm
    if not hasattr(cm, "get_cmap"):
        import matplotlib
        cm.get_cmap = lambda name=None, lut=None: matplotlib.colormaps[name or "viridis"]
    # GradLayer / attention_loss call .cuda() at construction / import time
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        nn.Module.cuda = lambda self, *a, **k: self
    from PIL import Image
    if not hasattr(Image, "ANTIALIAS"):
        Image.ANTIALIAS = Image.LANCZOS
    if not hasattr(np, "float"):
        np.float = float
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    """Returns a namespace with the reference symbols on the hot path."""
    import warnings
    install_stubs()
    warnings.filterwarnings("ignore")
    ns = types.SimpleNamespace()
    from packnet_code.packnet_sfm.networks.layers.packnet import layers01
    from packnet_code.packnet_sfm.networks.depth.PackNetSAN01 import PackNetSAN01
    from packnet_code.packnet_sfm.losses.grad_loss import GradLayer, GradLoss
    from packnet_code.packnet_sfm.losses.supervised_loss import SupervisedLoss, SilogLoss
    from packnet_code.packnet_sfm.models.SemiSupEdgeModel import SemiSupEdgeModel
    from packnet_code.packnet_sfm.utils.depth import inv2depth, depth2inv
    from packnet_code.packnet_sfm.utils.image import flip_lr, match_scales
    from packnet_code.packnet_sfm.models import model_utils
    ns.layers01 = layers01
    ns.PackNetSAN01 = PackNetSAN01
    ns.GradLayer, ns.GradLoss = GradLayer, GradLoss
    ns.SupervisedLoss, ns.SilogLoss = SupervisedLoss, SilogLoss
    ns.SemiSupEdgeModel = SemiSupEdgeModel
    ns.inv2depth, ns.depth2inv = inv2depth, depth2inv
    ns.flip_lr, ns.match_scales = flip_lr, match_scales
    ns.model_utils = model_utils
    return ns
