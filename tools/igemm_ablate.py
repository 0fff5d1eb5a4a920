"""development aid: main-loop ablation of the implicit-GEMM kernel (mte_debug_set(17, bits): leave out 1 MFMAs, 2 in-loop LDS-DMA,
4 fragment ds_reads).  Prints time per launch for each subset on a few layer shapes, 256x128 and 128x128 tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTE_USE_DEV_LIB", "1")      # development knobs live in libmte_hip_dev.so (-DMTE_DEV) only
import torch
from mindtheedge_amd import kernels as K

K.use_patch_kernels(False)
K.lib.mte_debug_set(15, 4)
names = {0: "full", 1: "-mfma", 2: "-dma", 4: "-dsread", 3: "dsread only", 5: "dma only", 6: "mfma only", 7: "loop skeleton"}
for big in (2, 1, 0):
    K.lib.mte_debug_set(6, big)
    print("tile:", ("128x128 (4 waves)", "256x128 (8 waves)", "256x256 (16 waves) where N >= 256, else 256x128")[big])
    K.lib.mte_debug_set(7, 100 if big == 2 else 224)
    for (B, H, W, cin, cout, k) in ((8, 24, 80, 512, 512, 3), (8, 48, 160, 256, 256, 3), (8, 96, 320, 128, 128, 3), (8, 48, 160, 512, 128, 5)):
        x = K.new_act(B, cin, H, W); x.normal_()
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        b = torch.zeros(cout, device="cuda")
        pack = K.WeightPack()
        wf, _ = pack.get(w, x.dtype, False)
        out = K.new_act(B, cout, H, W)
        K._splitk_workspace = lambda *a: (None, 0)
        fl = 2.0 * B * H * W * cin * cout * k * k
        line = []
        for col, abl in enumerate((0, 1, 2, 4, 3, 5, 6, 7, 0)):
            K.lib.mte_debug_set(17, abl)
            for _ in range(5):
                K.conv_forward(x, wf, b, cout, k, k, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                K.conv_forward(x, wf, b, cout, k, k, out=out)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 40 * 1e3
            line.append("%s %.1f us (%.0f TF)" % (names[abl], us, fl / us / 1e6))
        K.lib.mte_debug_set(17, 0)
        print("  M=%d Cin=%d N=%d k=%d: " % (B * H * W, cin, cout, k) + " | ".join(line))
