"""GPU (-m gpu): the LDS-tiled conv3d pack stencils must agree with the gather implementation on identical bf16 inputs
(same fp32 products, summation order differs only in the weight gradient)."""
import pytest
import torch

from conftest import rel_err, kernel_variant

pytestmark = pytest.mark.gpu


def _run(C, B, H, W, lds):
    from mindtheedge_amd import kernels as K
    K.set_compute_dtype("bf16")
    with kernel_variant(1, int(lds), 2):      # 0 gather kernels, 1 LDS-tiled (plane by plane), 2 + four-plane unpack backward [product]
        g = torch.Generator().manual_seed(C * 7 + H)
        x = K.image_to_act((torch.rand(B, C, H, W, generator=g) * 2 - 1).cuda()).detach().requires_grad_(True)
        w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda().requires_grad_(True)
        b3 = ((torch.rand(4, generator=g) - 0.5) * 0.4).cuda().requires_grad_(True)
        y = K.Pack3dFn.apply(x, w3, b3)
        G = (torch.rand(y.shape, generator=g) * 2 - 1).cuda()
        (y.float() * G).sum().backward()
        torch.cuda.synchronize()
        return y.float().cpu(), x.grad.float().cpu(), w3.grad.cpu(), b3.grad.cpu()


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 16, 32), (32, 1, 20, 36), (64, 1, 12, 40), (128, 1, 8, 16), (256, 1, 4, 16),
                                      (512, 1, 4, 8), (16, 2, 8, 12)])
def test_lds_pack3d_matches_gather(C, B, H, W):
    a = _run(C, B, H, W, True)
    r = _run(C, B, H, W, False)
    assert rel_err(a[0], r[0]) < 8e-3
    assert rel_err(a[1], r[1]) < 8e-3
    assert rel_err(a[2], r[2]) < 5e-4
    assert rel_err(a[3], r[3]) < 5e-4


def _run_unpack(C, B, H, W, lds):
    from mindtheedge_amd import kernels as K
    K.set_compute_dtype("bf16")
    with kernel_variant(1, int(lds), 2):      # 0 gather kernels, 1 LDS-tiled (plane by plane), 2 + four-plane unpack backward [product]
        g = torch.Generator().manual_seed(C * 3 + W)
        x = K.image_to_act((torch.rand(B, C, H, W, generator=g) * 2 - 1).cuda()).detach().requires_grad_(True)
        w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda().requires_grad_(True)
        b3 = ((torch.rand(4, generator=g) - 0.5) * 0.4).cuda().requires_grad_(True)
        y = K.Unpack3dFn.apply(x, w3, b3, None)
        G = (torch.rand(y.shape, generator=g) * 2 - 1).cuda()
        (y.float() * G).sum().backward()
        torch.cuda.synchronize()
        return y.float().cpu(), x.grad.float().cpu(), w3.grad.cpu(), b3.grad.cpu()


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 12, 20), (32, 1, 17, 33), (64, 1, 9, 40), (128, 1, 8, 16), (256, 1, 5, 16), (512, 1, 4, 8)])
def test_lds_unpack3d_backward_matches_gather(C, B, H, W):
    r = _run_unpack(C, B, H, W, 0)
    for mode in (1, 2):
        a = _run_unpack(C, B, H, W, mode)
        assert rel_err(a[0], r[0]) < 8e-3
        assert rel_err(a[1], r[1]) < 8e-3
        assert rel_err(a[2], r[2]) < 5e-4
        assert rel_err(a[3], r[3]) < 5e-4
