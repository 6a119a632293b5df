"""GPU: the fused GroupNorm (+ FiLM) (+ SiLU) kernels of the score networks (csrc/gn_act.hip) against torch's own ops,
forward and input gradient, and the score networks built on them against the reference-class fixtures."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('shape,groups,film,act,eps', [
    ((2, 64, 16, 16), 32, False, True, 1e-5), ((3, 128, 8, 8), 32, True, True, 1e-5), ((1, 32, 64, 64), 32, True, True, 1e-5),
    ((2, 96, 256), 32, False, False, 1e-5), ((2, 128, 32, 32), 32, False, True, 1e-6), ((4, 128, 256, 256), 32, True, True, 1e-5)])
def test_fused_group_norm_matches_torch(shape, groups, film, act, eps):
    from nhmc import unet
    g = torch.Generator().manual_seed(len(shape) * 100 + shape[1])
    C = shape[1]
    gn = torch.nn.GroupNorm(groups, C, eps=eps).cuda().requires_grad_(False)
    gn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g).cuda())
    gn.bias.copy_(0.1 * torch.randn(C, generator=g).cuda())
    x = (torch.randn(shape, generator=g) * 1.7 + 0.3).cuda()
    fm = (0.3 * torch.randn(shape[0], 2 * C, generator=g)).cuda() if film else None
    dy = torch.randn(shape, generator=g).cuda()

    def torch_form(xx):
        h = F.group_norm(xx, groups, gn.weight, gn.bias, eps)
        if fm is not None:
            sc, sh = fm.reshape(fm.shape + (1,) * (xx.dim() - 2)).chunk(2, dim=1)
            h = h * (1 + sc) + sh
        return F.silu(h) if act else h
    xa = x.clone().requires_grad_(True)
    ya = torch_form(xa.double()).float() if False else torch_form(xa)
    (ga,) = torch.autograd.grad(ya, xa, dy)
    xb = x.clone().requires_grad_(True)
    yb = unet.group_norm_act(gn, xb, act=act, film=fm)
    assert yb.grad_fn is not None and type(yb.grad_fn).__name__.startswith('_GroupNormAct')      # the fused path ran
    (gb,) = torch.autograd.grad(yb, xb, dy)
    # fp64 evaluation of the same formula as the common yardstick: the fused kernels are at least as close to it as ATen
    xd = x.double().requires_grad_(True)
    gnd = torch.nn.GroupNorm(groups, C, eps=eps).cuda().double()
    gnd.weight.data.copy_(gn.weight.double())
    gnd.bias.data.copy_(gn.bias.double())
    hd = gnd(xd)
    if fm is not None:
        sc, sh = fm.double().reshape(fm.shape + (1,) * (x.dim() - 2)).chunk(2, dim=1)
        hd = hd * (1 + sc) + sh
    yd = F.silu(hd) if act else hd
    (gd,) = torch.autograd.grad(yd, xd, dy.double())
    assert rel(yb, yd) < 2e-6 and rel(gb, gd) < 1e-5, (rel(yb, yd), rel(gb, gd))
    assert rel(yb, ya) < 2e-6 and rel(gb, ga) < 1e-5
    assert rel(yb, yd) <= 2 * rel(ya, yd) + 1e-7 and rel(gb, gd) <= 2 * rel(ga, gd) + 1e-6


def test_fallbacks_are_only_taken_where_documented():
    from nhmc import unet
    gn = torch.nn.GroupNorm(32, 64).cuda().requires_grad_(False)
    x = torch.randn(2, 64, 8, 8).cuda().requires_grad_(True)
    assert type(unet.group_norm_act(gn, x).grad_fn).__name__.startswith('_GroupNormAct')
    assert not type(unet.group_norm_act(gn.double(), x.double()).grad_fn).__name__.startswith('_GroupNormAct')   # float64 parity runs
    gn2 = torch.nn.GroupNorm(32, 64).cuda()                                                                       # trainable parameters
    assert not type(unet.group_norm_act(gn2, x).grad_fn).__name__.startswith('_GroupNormAct')
    os.environ['NHMC_FUSED_GN'] = '0'
    try:
        gn3 = torch.nn.GroupNorm(32, 64).cuda().requires_grad_(False)
        assert not type(unet.group_norm_act(gn3, x).grad_fn).__name__.startswith('_GroupNormAct')
    finally:
        os.environ.pop('NHMC_FUSED_GN')
    import nhmc._lib as L
    with pytest.raises(L.NhmcError):
        unet.K.gn_act_fwd(torch.randn(2, 64, 3, 3).cuda(), gn.weight, gn.bias, 32, 1e-5, 1)      # hw % 4 != 0 is refused, not emulated


@pytest.mark.parametrize('per_sample', [False, True])
def test_pre_bias_and_bias_residual_add(per_sample):
    """The producing convolution's bias (and, in the LDM blocks, the embedding term) entering the GroupNorm's load, and a
    convolution's bias folded into the residual add: same values and input gradients as the unfused torch sequence."""
    from nhmc import unet
    g = torch.Generator().manual_seed(5)
    B, C, H = 3, 64, 16
    gn = torch.nn.GroupNorm(32, C).cuda().requires_grad_(False)
    gn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g).cuda())
    gn.bias.copy_(0.1 * torch.randn(C, generator=g).cuda())
    x = torch.randn(B, C, H, H, generator=g).cuda()
    pre = (torch.randn(B, C, generator=g) if per_sample else torch.randn(C, generator=g)).cuda()
    dy = torch.randn(B, C, H, H, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    ya = F.silu(F.group_norm(xa + pre.reshape(((1, -1) if pre.dim() == 1 else tuple(pre.shape)) + (1, 1)), 32, gn.weight, gn.bias, gn.eps))
    (ga,) = torch.autograd.grad(ya, xa, dy)
    xb = x.clone().requires_grad_(True)
    yb = unet.group_norm_act(gn, xb, pre=pre)
    assert type(yb.grad_fn).__name__.startswith('_GroupNormAct')
    (gb,) = torch.autograd.grad(yb, xb, dy)
    assert rel(yb, ya) < 2e-6 and rel(gb, ga) < 1e-5
    # bias + residual add
    h = torch.randn(B, C, H, H, generator=g).cuda().requires_grad_(True)
    o = torch.randn(B, C, H, H, generator=g).cuda().requires_grad_(True)
    bias = torch.randn(C, generator=g).cuda()
    out = unet._BiasAdd2.apply(h, bias, o)
    assert torch.equal(out, (h + bias[None, :, None, None]) + o)
    gh, go = torch.autograd.grad(out, (h, o), dy)
    assert torch.equal(gh, dy) and torch.equal(go, dy)


def test_block_input_fork_adds_the_skip_gradient_inside_the_backward_kernel():
    """x feeds GroupNorm + SiLU and, unchanged, the skip path: the fork op returns an alias of x whose gradient is added
    inside the GroupNorm backward kernel -- the same bits as autograd's separate accumulation of the two gradients."""
    from nhmc import unet
    g = torch.Generator().manual_seed(8)
    B, C, H = 2, 64, 32
    gn = torch.nn.GroupNorm(32, C).cuda().requires_grad_(False)
    gn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g).cuda())
    x = torch.randn(B, C, H, H, generator=g).cuda()
    dy, ds = torch.randn(B, C, H, H, generator=g).cuda(), torch.randn(B, C, H, H, generator=g).cuda()
    xa = x.clone().requires_grad_(True)
    ya = unet.group_norm_act(gn, xa)
    (ga,) = torch.autograd.grad((ya * dy).sum() + (xa * ds).sum(), xa)                 # autograd adds the two gradients
    xb = x.clone().requires_grad_(True)
    yb, xs = unet.group_norm_act_fork(gn, xb)
    assert type(yb.grad_fn).__name__.startswith('_GroupNormActFork') and xs.data_ptr() == xb.data_ptr()
    (gb,) = torch.autograd.grad((yb * dy).sum() + (xs * ds).sum(), xb)
    assert torch.equal(yb, ya) and torch.equal(gb, ga)
    yc, _ = unet.group_norm_act_fork(gn, xb)                                            # skip path unused
    (gc,) = torch.autograd.grad((yc * dy).sum(), xb)
    (gd,) = torch.autograd.grad((unet.group_norm_act(gn, xa) * dy).sum(), xa)
    assert torch.equal(gc, gd)
    with torch.no_grad():
        yn, xn = unet.group_norm_act_fork(gn, x)
    assert xn is x and torch.equal(yn, ya)


def test_resblocks_with_and_without_the_fused_glue_agree():
    """A ResBlock of each network family (FFHQ scale-shift block, LDM additive-embedding block, VQ decoder block):
    fused glue vs NHMC_FUSED_GN=0, forward and input gradient."""
    from nhmc import ldm, unet
    torch.manual_seed(3)
    blocks = [(unet.ResBlock(64, 128, 96), True), (unet.ResBlock(64, 128, 64, down=True), True),
              (ldm.AddEmbResBlock(64, 128, 96), True), (ldm.PlainResBlock(64, 96), False)]
    for blk, takes_emb in blocks:
        blk = blk.cuda().eval().requires_grad_(False)
        x = torch.randn(2, 64, 16, 16).cuda()
        emb = torch.randn(2, 128).cuda()
        dy = None
        outs = []
        for mode in ('1', '0'):
            os.environ['NHMC_FUSED_GN'] = mode
            try:
                xl = x.clone().requires_grad_(True)
                y = blk(xl, emb) if takes_emb else blk(xl)
                dy = torch.randn_like(y) if dy is None else dy
                (gx,) = torch.autograd.grad(y, xl, dy)
                outs.append((y.detach(), gx))
            finally:
                os.environ.pop('NHMC_FUSED_GN')
        assert rel(outs[0][0], outs[1][0]) < 5e-6 and rel(outs[0][1], outs[1][1]) < 2e-5, type(blk).__name__


@pytest.mark.parametrize('up', [False, True])
@pytest.mark.parametrize('shape', [(2, 32, 16, 16), (3, 8, 64, 64), (1, 4, 256, 256)])
def test_resample_on_the_block_mean_kernels(up, shape):
    """unet.Resample (2x2 average pool / nearest 2x upsample) on the SR operator's kernels = torch's ops, forward bit for
    bit, backward too (each is the other's adjoint up to the factor 4)."""
    from nhmc import unet
    g = torch.Generator().manual_seed(shape[2] + up)
    x = torch.randn(shape, generator=g).cuda()
    r = unet.Resample(up)
    xa = x.clone().requires_grad_(True)
    ya = F.interpolate(xa, scale_factor=2, mode='nearest') if up else F.avg_pool2d(xa, 2)
    dy = torch.randn(ya.shape, generator=g).cuda()
    (ga,) = torch.autograd.grad(ya, xa, dy)
    xb = x.clone().requires_grad_(True)
    yb = r(xb)
    assert type(yb.grad_fn).__name__.startswith('_Resample2x')
    (gb,) = torch.autograd.grad(yb, xb, dy)
    assert torch.equal(yb, ya) and rel(gb, ga) < 1e-6
    assert not type(r(x.double().requires_grad_(True)).grad_fn).__name__.startswith('_Resample2x')     # float64 runs stay on torch
