#!/usr/bin/env python3
"""infer_edge_estimation.py -- compute core of the reference's depth-edge annotation driver
(/root/reference/infer_edge_estimation.py:119-259) on the MI355X: for one frame

    pred = model_wrapper.depth(image [, lidar / 200])['inv_depths'][0]            (:181-183, :216, :232)
    for every scale: probability = pred[scale] / 2 -> uint8 normals -> NMS -> hysteresis   (:186-206, :234-256)

with the network (RGB-only pass and, with a LiDAR map, the RGB+LiDAR pass through the sparse SAN branch) and the whole
post-processing on the device.  File handling of the reference (split files, PNG / .bin / .npy readers, cv2.imwrite, the
8-column output split list) is I/O plumbing and is not rebuilt: ``annotate_frame`` returns device tensors, ``--synthetic``
runs it on synthetic frames and prints what would be written.  The RGB+LiDAR pass uses the parity-unpinned SAN branch
(DESIGN.md 4.12); the post-processing is the pinned / Sobel-unpinned row f-2.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see mindtheedge_amd/__init__.py


def annotate_frame(model_wrapper, image, lidar_image=None, multiscale=True, nms=True, hysteresis=True, normals=True,
                   infer_rgb=True):
    """image: fp32 [1,3,H,W] in [0,1] on the GPU; lidar_image: fp32 [1,1,H,W] metres (zeros = no return) or None.
    -> {'regular': [(edges, normals), ...per scale], 'lidar': [...]}: edges float32 [1,H_s,W_s] (the reference writes
    edges * 255 as PNG), normals uint8 [1,H_s,W_s] or None.  Keys follow the reference's '_regular_00N' / '_lidar_00N' files."""
    import torch
    from mindtheedge_amd.utils.tools import annotate_edges
    scales = 4 if multiscale else 1
    model_wrapper.eval()
    out = {}
    with torch.no_grad():
        if infer_rgb:
            pred = model_wrapper.depth(image, rgb_edge=None)['inv_depths'][0]
            out['regular'] = annotate_edges(pred, nms=nms, hysteresis_=hysteresis, normals=normals, scales=scales)
        if lidar_image is not None:
            pred = model_wrapper.depth(image, lidar_image / 200.0, rgb_edge=None)['inv_depths'][0]      # reference :216 ("why 200???")
            out['lidar'] = annotate_edges(pred, nms=nms, hysteresis_=hysteresis, normals=normals, scales=scales)
    return out


def main():
    ap = argparse.ArgumentParser(description='depth-edge annotation (DEE inference + post-processing) on MI355X')
    ap.add_argument('--config', type=str, required=True, help='Input file (.yaml)')
    ap.add_argument('--synthetic', type=int, default=0, help='annotate N synthetic frames')
    ap.add_argument('--no-lidar', action='store_true')
    args = ap.parse_args()
    assert args.config.endswith('.yaml'), 'You need to provide a .yaml file'
    import time
    import torch
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    config = load_config(args.config, {'model': {'depth_net': {'with_san': not args.no_lidar}}})
    config.model.depth_net.checkpoint_path = config.model.depth_net.checkpoint_path if os.path.exists(
        config.model.depth_net.checkpoint_path or '') else ''
    K.set_compute_dtype('bf16')
    wrapper = ModelWrapper(config).cuda().eval()
    H, W = tuple(config.datasets.augmentation.image_shape) if not isinstance(config.datasets.augmentation.image_shape, str) \
        else eval(config.datasets.augmentation.image_shape)
    assert args.synthetic > 0, 'file readers are not part of this build: pass --synthetic N or call annotate_frame() on your tensors'
    g = torch.Generator(device='cuda').manual_seed(0)
    t0 = None
    for i in range(args.synthetic + 1):
        if i == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        image = torch.rand(1, 3, H, W, generator=g, device='cuda')
        lidar = None if args.no_lidar else (torch.rand(1, 1, H, W, generator=g, device='cuda') < 0.05).float() * \
            (2 + 70 * torch.rand(1, 1, H, W, generator=g, device='cuda'))
        out = annotate_frame(wrapper, image, lidar)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(args.synthetic, 1)
    for key, per_scale in out.items():
        for s, (e, n) in enumerate(per_scale):
            print('%08d_%s_%03d.png  edges %s kept %d   normals %s' % (args.synthetic - 1, key, s, tuple(e.shape), int((e > 0).sum()),
                                                                      None if n is None else tuple(n.shape)))
    print('%.2f ms per frame (%s passes, 4 scales, post-processing on device)' % (dt * 1e3, ' + '.join(out.keys())))


if __name__ == '__main__':
    main()
