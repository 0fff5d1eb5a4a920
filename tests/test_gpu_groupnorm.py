"""GPU (-m gpu): GroupNorm(16)+ELU kernels (csrc/norm_act.hip) through the C ABI against a torch-CPU statement of the
reference ops (nn.GroupNorm(16, C) + nn.ELU of Conv2D, layers01.py:32-38; the residual tail with Dropout2d factors,
layers01.py:62-73), forward and backward, for BOTH kernel families on the same inputs:

* single-pass slab kernels (a (sample, group) slab held in one workgroup's registers: the low-resolution layers),
* streaming two-pass kernels (everything else; forced with development knob 13 = 0),

with and without the second input / the fused conv-bias gradient, in fp32 mode (tight) and bf16 mode (storage rounding).
Shapes include the real ones: 512 channels at 24x80 (conv5 blocks), 256 at 48x160 (forward slab only), 512 at 12x40."""
import pytest
import torch
import torch.nn.functional as F

from conftest import kernel_variant

pytestmark = pytest.mark.gpu


def _reference(y1, y2, scale2, gamma, beta, dz):
    y1 = y1.double().requires_grad_(True)
    gamma, beta = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    v = y1
    if y2 is not None:
        y2 = y2.double().requires_grad_(True)
        v = y1 + y2 * scale2.double()[:, :, None, None]
    z = F.elu(F.group_norm(v, 16, gamma, beta, eps=1e-5))
    z.backward(dz.double())
    # bias gradient of the conv in front of the norm: column sums of d1 -- with a second input, of d2 (the shortcut conv's bias)
    return (z.detach(), y1.grad, None if y2 is None else y2.grad, gamma.grad, beta.grad, (y1.grad if y2 is None else y2.grad).sum(dim=(0, 2, 3)))


def _run(dtype, B, C, H, W, has2, slab):
    from mindtheedge_amd import kernels as K
    K.set_compute_dtype(dtype)
    tdt = K.compute_dtype()
    g = torch.Generator().manual_seed(B * 1000 + C + H + (7 if has2 else 0))
    rnd = lambda *s: torch.randn(*s, generator=g)
    # inputs are rounded to the compute dtype FIRST: both sides see identical values
    y1 = (rnd(B, C, H, W) * 1.5 + 0.3).to(tdt).float()
    y2 = (rnd(B, C, H, W)).to(tdt).float() if has2 else None
    scale2 = ((torch.rand(B, C, generator=g) >= 0.5).float() * 2.0) if has2 else None
    gamma, beta = 1.0 + 0.25 * rnd(C), 0.1 * rnd(C)
    dz = rnd(B, C, H, W).to(tdt).float()
    want = _reference(y1, y2, scale2, gamma, beta, dz)
    try:
        with kernel_variant(13, 1 if slab else 0, 1):
            dev = torch.device("cuda")
            a1 = K.as_act(y1.to(dev), tdt)
            a2 = K.as_act(y2.to(dev), tdt) if has2 else None
            sc = scale2.to(dev) if has2 else None
            gm, bt = gamma.to(dev), beta.to(dev)
            single = K.lib.mte_gn_fwd_is_single_pass_b(B, H * W, C, int(has2), K._dt(a1))
            z, stats = K._gn_forward(a1, a2, sc, gm, bt, 1e-5)
            out = K._gn_backward(K.as_act(dz.to(dev), tdt), a1, a2, sc, stats, gm, bt, 1e-5, has2, want_dbias=True)
            torch.cuda.synchronize()
            d1, d2, dgamma, dbeta = out[:4]
            dbias = out[4]
            got = (z.float().cpu(), d1.float().cpu(), None if d2 is None else d2.float().cpu(), dgamma.cpu(), dbeta.cpu(),
                   None if dbias is None else dbias.cpu())
            return got, want, single
    finally:
        K.set_compute_dtype("bf16")


def _err(a, b):
    b = b.double()
    return float((a.double() - b).abs().max() / b.abs().max().clamp(min=1e-30))


SHAPES = [(2, 512, 24, 80), (3, 512, 12, 40), (2, 256, 24, 40), (1, 128, 16, 24), (9, 256, 8, 16), (2, 64, 32, 64), (3, 32, 64, 64), (2, 128, 96, 320),
          (8, 512, 24, 80), (8, 256, 48, 160), (5, 128, 48, 160), (3, 256, 23, 79), (12, 512, 24, 80)]     # round 4: the cluster kernels' shapes (B = 8: 8 workgroups per slab; odd pixel counts; two sample sets)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("has2", [False, True])
@pytest.mark.parametrize("slab", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_groupnorm_elu_forward_backward(shape, slab, has2, dtype):
    B, C, H, W = shape
    got, want, single = _run(dtype, B, C, H, W, has2, slab)
    if not slab:
        assert single == 0
    tol = 2e-5 if dtype == "fp32" else 1.2e-2             # bf16: the OUTPUTS are rounded to 8 bits (2^-9 relative to their own size)
    gtol = 1e-4 if dtype == "fp32" else 2e-2
    names = ("z", "d1", "d2", "dgamma", "dbeta", "dbias")
    for n, a, b in zip(names, got, want):
        if a is None:
            continue
        t = tol if n in ("z", "d1", "d2") else gtol
        if n == "d2":
            b = want[2]
        assert _err(a, b) <= t, (n, _err(a, b), shape, slab, has2, dtype, single)


def test_slab_route_is_taken_for_the_low_resolution_layers():
    from mindtheedge_amd import kernels as K
    q = K.lib.mte_gn_fwd_is_single_pass
    assert q(12 * 40, 512, 0, 0) == 1 and q(12 * 40, 512, 1, 0) == 1        # unpack5 / iconv5 inputs (bf16): slab
    assert q(24 * 80, 512, 0, 0) == 0 and q(48 * 160, 256, 0, 0) == 0       # measured: streaming kernels win from 24x80 up
    assert q(96 * 320, 128, 0, 0) == 0 and q(192 * 640, 64, 0, 0) == 0 and q(384 * 1280, 32, 0, 0) == 0
    with kernel_variant(13, 0, 1):
        assert K.lib.mte_gn_fwd_is_single_pass(12 * 40, 512, 0, 0) == 0


def test_cluster_route_is_taken_and_is_bit_reproducible():
    """round 4: 512 channels at 24x80 and 256 at 48x160 (B = 8) run statistics + apply in ONE kernel, a cluster of workgroups per (sample,
    group); the partial sums are added in slot order, so two runs agree bit for bit -- also when the statistics buffer is reused without
    clearing (the last reader resets the exchange counters) -- and development knob 25 = 0 gives the streaming kernels' result to rounding."""
    from mindtheedge_amd import kernels as K
    q = K.lib.mte_gn_fwd_is_single_pass_b
    assert q(8, 24 * 80, 512, 0, 0) == 1 and q(8, 48 * 160, 256, 0, 0) == 1 and q(8, 24 * 80, 512, 1, 0) == 1
    assert q(8, 96 * 320, 128, 0, 0) == 0 and q(8, 192 * 640, 64, 0, 0) == 0 and q(8, 384 * 1280, 32, 0, 0) == 0
    K.set_compute_dtype("bf16")
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    for B, C, H, W in ((8, 512, 24, 80), (8, 256, 48, 160)):
        y = K.as_act((torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to(dev), torch.bfloat16)
        dz = K.as_act(torch.randn(B, C, H, W, generator=g).to(dev), torch.bfloat16)
        gm, bt = (1.0 + 0.25 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        outs = []
        for rep in range(4):
            z, stats = K._gn_forward(y, None, None, gm, bt, 1e-5)
            d1 = K._gn_backward(dz, y, None, None, stats, gm, bt, 1e-5, False)[0]
            torch.cuda.synchronize()
            outs.append((z.clone(), stats[:B * 32].clone(), d1.clone()))
        for o in outs[1:]:
            assert torch.equal(o[0], outs[0][0]), ("z", B, C, H, W)
            assert torch.equal(o[1], outs[0][1]), ("stats", B, C, H, W)
            if C == 512:                                         # (256 @48x160: the backward stays on the streaming kernels, whose reduce pass uses float atomics)
                assert torch.equal(o[2], outs[0][2]), ("d1", B, C, H, W)
        # the same statistics buffer twice, not cleared in between
        z1, stats = K._gn_forward(y, None, None, gm, bt, 1e-5)
        K.lib.mte_gn_elu_fwd(K._pl(y)[0], K._pl(y)[1], 0, 0, 0, stats.data_ptr(), 0, gm.data_ptr(), bt.data_ptr(), K._pl(z1)[0], K._pl(z1)[1],
                             B, H * W, C, 1e-5, K._dt(y), K._stream())
        torch.cuda.synchronize()
        assert torch.equal(z1, outs[0][0])
        with kernel_variant(25, 0, 1):
            zs, st2 = K._gn_forward(y, None, None, gm, bt, 1e-5)
            ds = K._gn_backward(dz, y, None, None, st2, gm, bt, 1e-5, False)[0]
            torch.cuda.synchronize()
        assert float((zs.float() - outs[0][0].float()).abs().max()) <= 2.0 ** -6 * float(zs.float().abs().max())
        assert float((ds.float() - outs[0][2].float()).abs().max()) <= 2.0 ** -6 * float(ds.float().abs().max())
