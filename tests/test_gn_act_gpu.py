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
