"""development aid: forward implicit-GEMM time vs K depth for one output shape (fixed per-tile overhead vs per-K-step cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTE_USE_DEV_LIB", "1")      # development knobs live in libmte_hip_dev.so (-DMTE_DEV) only
import torch
from mindtheedge_amd import kernels as K

K.use_patch_kernels(False)
for knob in (1, 2):
  K.lib.mte_debug_set(6, knob)
  print("big tiles:", knob)
  for (B, H, W, cout) in ((8, 96, 320, 128), (8, 48, 160, 256), (8, 24, 80, 512)):
      for cin in (128, 256, 512):
          x = K.new_act(B, cin, H, W); x.normal_()
          w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
          b = torch.zeros(cout, device="cuda")
          pack = K.WeightPack()
          wf, _ = pack.get(w, x.dtype, False)
          out = K.new_act(B, cout, H, W)
          orig = K._splitk_workspace
          K._splitk_workspace = lambda *a: (None, 0)
          for _ in range(3):
              K.conv_forward(x, wf, b, cout, 3, 3, out=out)
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(20):
              K.conv_forward(x, wf, b, cout, 3, 3, out=out)
          e1.record(); torch.cuda.synchronize()
          K._splitk_workspace = orig
          us = e0.elapsed_time(e1) / 20 * 1e3
          fl = 2.0 * B * H * W * cin * cout * 9
          tiles = (B * H * W // 128) * ((cout + 127) // 128)
          print("M=%d N=%d Cin=%4d ksteps=%4d tiles=%d: %7.1f us  %6.1f TF/s" % (B * H * W, cout, cin, 9 * cin // 32, tiles, us, fl / us / 1e6))
