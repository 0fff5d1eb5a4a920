#!/usr/bin/env python3
"""infer_edges.py -- depth inference core of the reference script (/root/reference/infer_edges.py:237-366, rows H3):
image file -> PIL + LANCZOS resize -> [1,3,H,W] in [0,1] -> model_wrapper.depth(image)['inv_depths'][0][0] -> inv2depth ->
``NNNNNNNN_regular.npy`` (float32 metres) + ``NNNNNNNN_regular.png`` (depth / max * 255).
The reference's BSDS / Canny evaluation around it is out of scope (py-bsds500, OpenCV)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see mindtheedge_amd/__init__.py: side stream / RCCL streams need their own hardware queues


def infer_depth(model_wrapper, image):
    """image: fp32 [B,3,H,W] on the GPU, H and W multiples of 32 -> depth [B,1,H,W] fp32 (metres)."""
    import torch
    from mindtheedge_amd.utils.depth import inv2depth
    from mindtheedge_amd import kernels as K
    model_wrapper.eval()
    with torch.no_grad():
        pred_inv_depth = model_wrapper.depth(image, rgb_edge=None)['inv_depths'][0][0]
    depth = inv2depth(pred_inv_depth)
    # The eval forward runs the GroupNorm cluster kernels too (bounded inter-workgroup waits); the optimizer that polls the device error word on the
    # training path never runs here.  The depth goes to the host next (save_depth), so waiting for the stream costs nothing that is not paid anyway.
    torch.cuda.current_stream().synchronize()
    K.check_device_errors()
    return depth


IMAGE_EXT = ('.png', '.jpg', '.jpeg', '.bmp', '.ppm')


def load_frame(path, image_shape=None):
    """One input frame -> float32 [3,H,W] in [0,1] on the host, the way the reference prepares it (infer_edges.py:266-282):
    PIL ``Image.open`` -> ``resize_image`` to the configured (H, W) with LANCZOS (datasets/augmentations.py:16-35; ANTIALIAS is
    LANCZOS) -> ``to_tensor`` (HWC uint8 / 255 -> CHW).  ``.npy`` arrays ([3,H,W] or [H,W,3], [0,1] or uint8) are taken as they are."""
    import numpy as np
    import torch
    if path.lower().endswith(IMAGE_EXT):
        from PIL import Image
        im = Image.open(path).convert('RGB')
        if image_shape is not None and tuple(image_shape) != (im.size[1], im.size[0]):
            im = im.resize((int(image_shape[1]), int(image_shape[0])), Image.LANCZOS)
        a = np.asarray(im, dtype=np.uint8)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1), dtype=np.float32) / 255.0)
    a = np.load(path).astype(np.float32)
    if a.shape[-1] == 3:
        a = a.transpose(2, 0, 1)
    return torch.from_numpy(np.ascontiguousarray(a) / np.float32(255.0 if a.max() > 1.5 else 1.0))


def save_depth(output_base, depth, png=True):
    """``<base>_regular.npy`` (float32 metres) and, as the reference (infer_edges.py:349-353), ``<base>_regular.png`` =
    depth / max * 255 as 8-bit grey (cv2.imwrite of a float array saturates-and-rounds to uint8)."""
    import numpy as np
    d = depth.detach().float().cpu().numpy()
    np.save(output_base + '_regular.npy', d)
    if png:
        from PIL import Image
        Image.fromarray(np.clip(np.rint(d / d.max() * 255.0), 0, 255).astype(np.uint8)).save(output_base + '_regular.png')


def main():
    ap = argparse.ArgumentParser(description='PackNet-SAN depth inference on MI355X')
    ap.add_argument('--config', type=str, required=True, help='YAML (model.depth_net.checkpoint_path may name a .ckpt)')
    ap.add_argument('--input', type=str, nargs='*', default=[], help='image files (png / jpg: resized to the configured shape with LANCZOS, as the '
                    'reference) or .npy arrays [3,H,W] / [H,W,3] in [0,1] (or uint8)')
    ap.add_argument('--no-png', action='store_true', help='write only the .npy depth maps')
    ap.add_argument('--output', type=str, default='results')
    ap.add_argument('--synthetic', type=int, default=0, help='number of synthetic frames of the configured shape')
    ap.add_argument('--graph', action='store_true', help='replay the forward from a HIP graph (single frames are launch-bound)')
    args = ap.parse_args()
    import numpy as np
    import torch
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    config = load_config(args.config)
    if not os.path.exists(config.model.depth_net.checkpoint_path or ''):
        config.model.depth_net.checkpoint_path = ''
    wrapper = ModelWrapper(config).cuda()
    os.makedirs(args.output, exist_ok=True)
    shape = config.datasets.augmentation.image_shape
    import ast
    H, W = ast.literal_eval(shape) if isinstance(shape, str) else tuple(shape)
    images = [load_frame(p, (H, W)) for p in args.input]
    g = torch.Generator().manual_seed(0)
    images += [torch.rand(3, H, W, generator=g) for _ in range(args.synthetic)]
    graphed = None
    for ctr, img in enumerate(images):
        frame = img.unsqueeze(0).cuda()
        if args.graph:
            from mindtheedge_amd.utils.depth import inv2depth
            from mindtheedge_amd.utils.graph import GraphedDepth
            if graphed is None or tuple(graphed.rgb.shape) != tuple(frame.shape):
                wrapper.eval()
                graphed = GraphedDepth(wrapper.depth_net, frame)
            depth = inv2depth(graphed(frame)['inv_depths'][0][0])
        else:
            depth = infer_depth(wrapper, frame)
        save_depth(os.path.join(args.output, '%08d' % ctr), depth[0, 0], png=not args.no_png)
    print('wrote %d depth maps to %s' % (len(images), args.output))


if __name__ == '__main__':
    main()
