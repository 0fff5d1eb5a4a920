"""Synthetic batches with the training batch schema of SURVEY.md 8(a)/(d) (what GTADataset.__getitem__ +
train_transforms + default collate deliver; 'input_depth' only with lidar=True): generated on the device."""
import math

import torch


def synthetic_batch(B, H, W, seed, device, lidar=False):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, device=device)
    batch = {'rgb': r(B, 3, H, W)}
    batch['depth'] = (r(B, 1, H, W) < 0.05).float() * (1.0 + 79.0 * r(B, 1, H, W))
    for s in range(4):
        sfx = '' if s == 0 else '_%d' % s
        h, w = H >> s, W >> s
        batch['edge' + sfx] = (r(B, 1, h, w) < 0.03).float() * r(B, 1, h, w)
        batch['normal' + sfx] = (r(B, 1, h, w) * 2 - 1) * math.pi
    if lidar:        # sparse LiDAR input of the RGB+LiDAR pass (gta_dataset.py:380-382): a 5 % sample of a dense depth map, metres
        batch['input_depth'] = (r(B, 1, H, W) < 0.05).float() * (1.0 + 79.0 * r(B, 1, H, W))
    return batch


class SyntheticLoader:
    """Iterable of `steps` device-resident batches; rank-decorrelated seeds (DistributedSampler analogue)."""

    def __init__(self, B, H, W, steps, device, rank=0, pool=2, lidar=False):
        self.batches = [synthetic_batch(B, H, W, 1234 + 97 * rank + i, device, lidar=lidar) for i in range(pool)]
        self.steps = steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % len(self.batches)]

    def __len__(self):
        return self.steps
