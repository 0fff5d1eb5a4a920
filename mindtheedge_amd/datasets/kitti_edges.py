"""Data formats of the depth-edge training set (data half) -- the pieces of the reference's
packnet_sfm/datasets/gta_dataset.py and augmentations.py that define WHAT the training batch contains:

  * the 8-column split file (gta_dataset.py:184-211): image, depth, edge, lidar, seg, rgb_edge, rgb_edge_for_loss, normal
  * multi-scale annotation naming (gta_dataset.py:366-369,414-418): ``..._000.png`` plus ``_001`` .. ``_003`` for the
    half / quarter / eighth resolution edge and normal maps
  * target preparation, on the device: edges / 255, normals de-quantised from uint8 to radians, sparse maps resized with
    ``resize_depth_preserve`` (augmentations.py:58-100,159-217)

File decoding (PNG / .npy readers) stays on the host with PIL / numpy, as upstream; everything arithmetic runs in the
HIP library (csrc/data_prep.hip) on CUDA tensors -- host tensors raise MteError.
"""
import os

import torch

SPLIT_COLUMNS = ('rgb', 'depth', 'edge', 'lidar', 'seg', 'rgb_edge', 'rgb_edge_for_loss', 'normal')


def parse_split_line(line):
    """One line of the split file -> {column: path or None} following gta_dataset.py:184-211 (space separated, a trailing
    newline column is dropped, the literal 'None' marks an absent seg / rgb_edge / normal file)."""
    names = line.split(' ')
    if names[-1] == '\n':
        names = names[:-1]
    rec = {}
    for col, name in zip(SPLIT_COLUMNS, names):
        path = name.split('\n')[0]
        rec[col] = None if path in ('None', '') else path
    return rec


def read_split(path):
    with open(path, 'r') as f:
        return [parse_split_line(line) for line in f.readlines() if line.strip()]


def multiscale_paths(path_000, require_existing=True):
    """[scale 0, 1, 2, 3] file names of an edge / normal annotation (gta_dataset.py:366-369: the stem before '_000' plus
    '_00N.png'); with require_existing only scale 0 is returned when '_001.png' is not on disk, like the reference."""
    stem = path_000.split('_000')[0]
    if require_existing and not os.path.exists(stem + '_001.png'):
        return [path_000]
    return [path_000] + [stem + '_00' + str(i) + '.png' for i in range(1, 4)]


def _u8(t):
    from .. import kernels as K
    K._require_gpu(t)
    if t.dtype != torch.uint8:
        raise ValueError("expected the uint8 map read from the PNG, got {}".format(t.dtype))
    return t.contiguous()


def edge_target(edge_u8):
    """uint8 edge annotation -> float32 target: /255 when the map is on the 0..255 scale (max > 1), unchanged otherwise
    (augmentations.py:186-188).  One host read (the maximum) per map, as upstream."""
    from .. import kernels as K
    src = _u8(edge_u8)
    if not bool(src.max() > 1):
        return src.float()
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    K.lib.mte_edge_target_from_u8(src.data_ptr(), out.data_ptr(), src.numel(), K._stream())
    return out


def normal_target(normal_u8):
    """uint8 normal annotation -> radians: (360 * (v / 255) - 180) * pi / 180 (gta_dataset.py:407-409), float64 inside."""
    from .. import kernels as K
    src = _u8(normal_u8)
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    K.lib.mte_normal_target_from_u8(src.data_ptr(), out.data_ptr(), src.numel(), K._stream())
    return out


def resize_depth_preserve(depth, shape):
    """augmentations.py:58-100 for float maps [h,w] / [B,h,w] on the device -> [H,W] / [B,H,W]: every valid (> 0) pixel
    moves to (int(y*H/h), int(x*W/w)); the last one in raster order wins a shared target; zeros elsewhere."""
    from .. import kernels as K
    K._require_gpu(depth)
    squeeze = depth.dim() == 2
    src = (depth.unsqueeze(0) if squeeze else depth).detach().float().contiguous()
    if src.dim() != 3:
        raise ValueError("expected [h,w] or [B,h,w], got {}".format(tuple(depth.shape)))
    if not isinstance(shape, (tuple, list)):
        shape = tuple(int(s * shape) for s in src.shape[-2:])               # a single number is a resize ratio (:78-79)
    B, h, w = src.shape
    H, W = int(shape[0]), int(shape[1])
    out = torch.empty((B, H, W), dtype=torch.float32, device=src.device)
    ws = torch.empty((B, H, W), dtype=torch.int32, device=src.device)
    K.lib.mte_resize_depth_preserve(src.data_ptr(), B, h, w, out.data_ptr(), H, W, ws.data_ptr(), K._stream())
    return out[0] if squeeze else out


def prepare_edge_sample(edge_maps_u8, normal_maps_u8, shape):
    """The edge / normal part of resize_sample (augmentations.py:178-211) for one sample: lists of uint8 maps for scales
    0..3 (CUDA tensors [h_s,w_s]) -> {'edge', 'edge_1'.., 'normal', 'normal_1'..} float32 [1,H_s,W_s] with H_s = H >> s.
    Normals keep their size when it already is the target size; otherwise they are resized bilinearly like cv2.resize
    (parity-unpinned restatement, utils/edge.py::resize_linear)."""
    from ..utils.edge import resize_linear
    H, W = int(shape[0]), int(shape[1])
    out = {}
    for s, e in enumerate(edge_maps_u8):
        key = 'edge' if s == 0 else 'edge_%d' % s
        tgt = (int(H / (2 ** s)), int(W / (2 ** s)))
        resized = resize_depth_preserve(_u8(e).float(), tgt)
        out[key] = (resized / 255.0 if bool(resized.max() > 1) else resized).unsqueeze(0)
    for s, n in enumerate(normal_maps_u8):
        key = 'normal' if s == 0 else 'normal_%d' % s
        tgt = (int(H / (2 ** s)), int(W / (2 ** s)))
        rad = normal_target(n)
        out[key] = (rad if tuple(rad.shape) == tgt else resize_linear(rad, tgt)).unsqueeze(0)
    return out


# ---- a reader on top of the formats above -------------------------------------------------------------------------------
# File decoding is host I/O (PIL / numpy, the reference uses PIL and cv2.imread); everything after it -- sparse resizes,
# target scaling, de-quantisation -- runs on the device through the functions above.  Not rebuilt: colour jitter, random
# crops, context frames, the .bin velodyne projection (pass depth / lidar maps as 16-bit PNG or .npy).

def _read_gray_u8(path):
    """cv2.imread(path)[:, :, 0] of the reference for the 8-bit single-channel annotation PNGs."""
    import numpy as np
    from PIL import Image
    if path.endswith('.npy'):
        return np.load(path)
    a = np.array(Image.open(path))
    return a if a.ndim == 2 else a[:, :, -1 if a.shape[2] >= 3 else 0]      # OpenCV channel 0 is blue = PIL's channel 2


def read_png_depth(path):
    """KITTI 16-bit depth PNG -> float32 metres, -1 where there is no return (reference kitti_dataset.py:40-46)."""
    import numpy as np
    from PIL import Image
    if path.endswith('.npy'):
        return np.load(path).astype(np.float32)
    png = np.array(Image.open(path), dtype=int)
    assert png.max() > 255, 'Wrong .png depth file'
    depth = png.astype(np.float32) / 256.
    depth[png == 0] = -1.
    return depth


class KittiEdgeSplitDataset:
    """One sample per split-file line: rgb (PIL-resized to ``image_shape`` like the reference's resize_image, /255),
    depth (resize_depth_preserve), edge / edge_1..3 and normal / normal_1..3 targets.  ``__getitem__`` returns device
    tensors shaped like the reference's collated sample with batch size 1 removed: rgb [3,H,W], depth [1,H,W], ..."""

    def __init__(self, split_file, image_shape, device='cuda', root=''):
        self.records = read_split(split_file)
        self.shape = (int(image_shape[0]), int(image_shape[1]))
        self.device = torch.device(device)
        self.root = root

    def __len__(self):
        return len(self.records)

    def _p(self, path):
        return path if os.path.isabs(path) or not self.root else os.path.join(self.root, path)

    def __getitem__(self, idx):
        import numpy as np
        from PIL import Image
        rec = self.records[idx]
        H, W = self.shape
        img = Image.open(self._p(rec['rgb'])).convert('RGB')
        if img.size != (W, H):
            img = img.resize((W, H), Image.LANCZOS)                       # transforms.Resize(shape, ANTIALIAS), augmentations.py:16-35
        rgb = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).to(self.device).permute(2, 0, 1).float() / 255.0
        sample = {'idx': idx, 'rgb': rgb}
        if rec.get('depth'):
            d = torch.from_numpy(read_png_depth(self._p(rec['depth']))).to(self.device)
            sample['depth'] = resize_depth_preserve(d, self.shape).unsqueeze(0)
        edges, normals = [], []
        if rec.get('edge'):
            edges = [torch.from_numpy(np.ascontiguousarray(_read_gray_u8(p)).astype(np.uint8)).to(self.device)
                     for p in multiscale_paths(self._p(rec['edge']))]
        if rec.get('normal'):
            normals = [torch.from_numpy(np.ascontiguousarray(_read_gray_u8(p)).astype(np.uint8)).to(self.device)
                       for p in multiscale_paths(self._p(rec['normal']))]
        sample.update(prepare_edge_sample(edges, normals, self.shape))
        return sample


def collate(samples):
    out = {}
    for k in samples[0]:
        vals = [s[k] for s in samples]
        out[k] = torch.stack(vals) if torch.is_tensor(vals[0]) else vals
    return out


class SplitLoader:
    """Minimal epoch iterator: rank-strided, fixed order or shuffled per epoch (``set_epoch``), drops the ragged tail."""

    def __init__(self, dataset, batch_size, rank=0, world=1, shuffle=True, seed=0):
        self.ds, self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch = dataset, batch_size, rank, world, shuffle, seed, 0
        self.sampler = self                                           # Trainer.fit calls loader.sampler.set_epoch(epoch)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return len(self.ds) // self.world // self.bs

    def __iter__(self):
        order = torch.randperm(len(self.ds), generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist() \
            if self.shuffle else list(range(len(self.ds)))
        mine = order[self.rank::self.world]
        for i in range(len(self)):
            yield collate([self.ds[j] for j in mine[i * self.bs:(i + 1) * self.bs]])


def make_loader(config, rank, world):
    """``train_edges.py --data mindtheedge_amd.datasets.kitti_edges:make_loader``: config.datasets.train.split[0] is the
    8-column split file, config.datasets.train.path[0] (optional) the root folder of its relative paths."""
    tr = config.datasets.train
    shape = config.datasets.augmentation.image_shape
    shape = eval(shape) if isinstance(shape, str) else tuple(shape)
    root = (tr.get('path') or [''])[0] if isinstance(tr.get('path'), (list, tuple)) else (tr.get('path') or '')
    split = tr['split'][0] if isinstance(tr['split'], (list, tuple)) else tr['split']
    ds = KittiEdgeSplitDataset(split, shape, device=torch.device('cuda', torch.cuda.current_device()), root=root)
    return SplitLoader(ds, int(tr.batch_size), rank, world)
