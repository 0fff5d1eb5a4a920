"""GPU (-m gpu): the LDS-tiled conv3d pack stencils must agree with the gather implementation on identical bf16 inputs
(same fp32 products, summation order differs only in the weight gradient)."""
import pytest
import torch

from conftest import rel_err, kernel_variant

pytestmark = pytest.mark.gpu
PRODUCT = 539        # development knob 1 = 300 + g_p3_mfma_data of the product build (csrc/pack3d.hip)


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


def _unpack_bwd_data(C, B, H, W, knob, seed):
    """dx of the unpack layer's conv3d for one kernel variant (development knob 1: 300 = fp32-VALU stencil, 301 = matrix cores with interleaved
    planes, 303 / 307 = + raw records by LDS-DMA for C = 32 with 2 / 4 waves; + 8 = forward as banded GEMM, + 32 = unpack forward with the spatial taps in K, + 64 = pack forward in that form [539 = product: + 128 = pack backward data on the matrix cores])"""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd._lib import dev_library
    K.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(seed)
    dout = K.image_to_act((torch.rand(B, C, 2 * H, 2 * W, generator=g) * 2 - 1).cuda())
    w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda()
    dx = K.new_act(B, C, H, W)
    dx.fill_(7.0)
    dp, ldo = K._pl(dout)
    xp, ldx = K._pl(dx)
    with dev_library() as lib:
        lib.mte_debug_set(1, knob)
        try:
            for _ in range(2):
                lib.mte_unpack3d_bwd_data(dp, ldo, w3.data_ptr(), xp, ldx, B, H, W, C, K._dt(dx), K._stream())
            torch.cuda.synchronize()
        finally:
            lib.mte_debug_set(1, PRODUCT)
    return dx.float().cpu()


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 16, 32), (32, 1, 17, 33), (32, 3, 8, 16), (32, 1, 1, 1), (32, 1, 9, 47), (64, 1, 12, 48), (64, 2, 5, 19),
                                      (64, 1, 4, 16), (32, 2, 96, 160)])
def test_matrix_core_unpack3d_backward_data_matches_the_valu_stencil(C, B, H, W):
    """banded-operand MFMA forms (weights as bf16 hi + lo parts: ~fp32 weights) against the fp32-VALU stencil on the same bf16 gradient:
    the results may differ by one bf16 rounding of the output where the fp32 sums differ in the last bits"""
    ref = _unpack_bwd_data(C, B, H, W, 300, seed=C + H)
    for knob in (301, 303, 307):
        got = _unpack_bwd_data(C, B, H, W, knob, seed=C + H)
        d = (got - ref).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(ref.abs().max()), (knob, float(d.max()))         # one bf16 ulp at the largest magnitude
        assert float((d > 0).float().mean()) < 0.05, (knob, float((d > 0).float().mean()))           # ... and rarely
        rms = float(d.double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
        assert rms < 1e-3, (knob, rms)


def _unpack_fwd(C, B, H, W, knob, seed, persist=1024):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd._lib import dev_library
    K.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(seed)
    x = K.image_to_act((torch.rand(B, C, H, W, generator=g) * 2 - 1).cuda())
    w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda()
    b3 = ((torch.rand(4, generator=g) - 0.5) * 0.4).cuda()
    out = K.new_act(B, C, 2 * H, 2 * W)
    out.fill_(7.0)
    xp, ldx = K._pl(x)
    op, ldo = K._pl(out)
    with dev_library() as lib:
        lib.mte_debug_set(1, knob)
        lib.mte_debug_set(1, 2000 + persist)          # workgroups of the persistent kernel: few -> every workgroup walks many tiles (double-buffered LDS)
        try:
            for _ in range(2):
                lib.mte_unpack3d_fwd(xp, ldx, w3.data_ptr(), b3.data_ptr(), op, ldo, B, H, W, C, K._dt(x), K._stream())
            torch.cuda.synchronize()
        finally:
            lib.mte_debug_set(1, PRODUCT)
            lib.mte_debug_set(1, 2000 + 1024)
    return out.float().cpu()


def _close_to_valu(got, ref):
    d = (got - ref).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(ref.abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.05, float((d > 0).float().mean())
    rms = float(d.double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
    assert rms < 1e-3, rms


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 16, 32), (32, 1, 17, 33), (32, 3, 8, 16), (32, 1, 1, 1), (32, 1, 9, 47), (64, 1, 12, 48), (64, 2, 5, 19),
                                      (64, 1, 4, 16), (32, 2, 96, 160), (128, 1, 7, 21), (128, 2, 8, 16), (256, 1, 5, 16), (256, 2, 3, 33)])
def test_matrix_core_unpack3d_forward_matches_the_valu_stencil(C, B, H, W):
    """conv3d(1 -> 4) + pixel shuffle on the same bf16 input: the fp32-VALU gather kernel (307) against the banded-operand MFMA form (315, C <= 64)
    and the taps-in-K form with transposing LDS reads (539 = product, C <= 256)"""
    ref = _unpack_fwd(C, B, H, W, 307, seed=C + W)
    _close_to_valu(_unpack_fwd(C, B, H, W, PRODUCT, seed=C + W), ref)
    if C <= 64:
        got = _unpack_fwd(C, B, H, W, 315, seed=C + W)
        for wgs in (8, 24):                                # the same tiles dealt to 8 / 24 workgroups: bit-identical
            assert torch.equal(_unpack_fwd(C, B, H, W, 315, seed=C + W, persist=wgs), got), wgs
        _close_to_valu(got, ref)


def _pack_fwd(C, B, H, W, knob, seed):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd._lib import dev_library
    K.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(seed)
    x = K.image_to_act((torch.rand(B, C, H, W, generator=g) * 2 - 1).cuda())
    w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda()
    b3 = ((torch.rand(4, generator=g) - 0.5) * 0.4).cuda()
    out = K.new_act(B, 16 * C, H // 2, W // 2)
    out.fill_(7.0)
    xp, ldx = K._pl(x)
    op, ldo = K._pl(out)
    with dev_library() as lib:
        lib.mte_debug_set(1, knob)
        try:
            for _ in range(2):
                lib.mte_pack3d_fwd(xp, ldx, w3.data_ptr(), b3.data_ptr(), op, ldo, B, H, W, C, K._dt(x), K._stream())
            torch.cuda.synchronize()
        finally:
            lib.mte_debug_set(1, PRODUCT)
    return out.float().cpu()


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 16, 32), (32, 1, 6, 70), (64, 1, 12, 40), (64, 2, 10, 6), (128, 1, 8, 36), (256, 1, 4, 34), (512, 1, 4, 8), (64, 16, 6, 64)])
def test_matrix_core_pack3d_forward_matches_the_valu_stencil(C, B, H, W):
    """space-to-depth + conv3d(1 -> 4): the taps-in-K MFMA form (depth slabs of 128; 539 = product) against the fp32-VALU LDS stencil (347)"""
    _close_to_valu(_pack_fwd(C, B, H, W, PRODUCT, seed=C + H), _pack_fwd(C, B, H, W, 347, seed=C + H))


def _pack_bwd_data(C, B, H, W, knob, seed):
    """dx of the pack layer's conv3d for one kernel variant (knob 1: 411 = fp32-VALU LDS stencil, 539 = matrix cores [product], 555 = ... with the weights as one bf16 value)"""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd._lib import dev_library
    K.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(seed)
    dout = K.image_to_act((torch.rand(B, 16 * C, H // 2, W // 2, generator=g) * 2 - 1).cuda())
    w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda()
    dx = K.new_act(B, C, H, W)
    dx.fill_(7.0)
    dp, ldo = K._pl(dout)
    xp, ldx = K._pl(dx)
    with dev_library() as lib:
        lib.mte_debug_set(1, knob)
        try:
            for _ in range(2):
                lib.mte_pack3d_bwd_data(dp, ldo, w3.data_ptr(), xp, ldx, B, H, W, C, K._dt(dx), K._stream())
            torch.cuda.synchronize()
        finally:
            lib.mte_debug_set(1, PRODUCT)
    return dx.float().cpu()


@pytest.mark.parametrize("C,B,H,W", [(32, 2, 16, 32), (32, 1, 18, 34), (32, 3, 8, 64), (32, 1, 2, 2), (32, 1, 10, 94), (64, 1, 12, 48), (64, 2, 10, 38),
                                      (128, 1, 8, 36), (256, 1, 4, 34), (512, 1, 4, 8), (32, 2, 96, 160), (64, 16, 6, 64)])
def test_matrix_core_pack3d_backward_data_matches_the_valu_stencil(C, B, H, W):
    """un-shuffle(conv3d^T(dO)) on the same bf16 gradient: the banded-operand MFMA form of round 6 (whole tiles, ragged tiles, one-pixel images, 1 .. 16 depth
    segments) against the fp32-VALU LDS stencil; every element of dx is written"""
    ref = _pack_bwd_data(C, B, H, W, 411, seed=C + H)
    _close_to_valu(_pack_bwd_data(C, B, H, W, 539, seed=C + H), ref)
    one = _pack_bwd_data(C, B, H, W, 555, seed=C + H)                # weights rounded to bf16: the conv3d weights' 2^-9 relative error
    rms = float((one - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
    assert rms < 6e-3, rms
