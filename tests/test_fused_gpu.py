"""GPU: the fused (inpainting data term + last DDIM-step VJP) kernel equals the two-kernel path bit for bit."""
import copy

import pytest
import torch

from oracle import operators as oops, schedule as osched

pytestmark = pytest.mark.gpu
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


@pytest.mark.parametrize('dim,B', [(16, 1), (32, 3), (256, 2)])
def test_fused_last_vjp_equals_two_kernel_path(dim, B):
    import nhmc.kernels as K
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim)
    op = operators.Inpainting(3, dim, oops.random_inpaint_missing(dim, generator=g_), 'cuda')
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    loss_a, g = op.data_term(cur, y, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b)
    assert float((loss_a - loss_b).abs().max() / loss_a.abs().max()) < 1e-12


def test_engine_with_and_without_fusion_agree(tiny_score):
    from nhmc import operators, plugin, sampler
    dim, B = 32, 3
    g_ = torch.Generator().manual_seed(9)
    op = operators.Inpainting(3, dim, oops.random_inpaint_missing(dim, generator=g_), 'cuda')
    algo = plugin.HMC(copy.deepcopy(tiny_score).cuda(), op, 0.1)
    eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, torch.device('cuda'))
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    a = eng.decode_and_grad(x, y)
    eng.fuse_last = False
    b = eng.decode_and_grad(x, y)
    for u, v in zip(a, b):
        assert torch.equal(u, v) or float((u - v).abs().max()) <= 1e-12 * float(v.abs().max())


@pytest.mark.parametrize('dim,B,ratio', [(16, 1, 2), (32, 3, 4), (64, 2, 8), (48, 2, 16), (256, 2, 4), (256, 2, 16)])
def test_fused_sr_last_vjp_equals_two_kernel_path(dim, B, ratio):
    import nhmc.kernels as K
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim + ratio)
    op = operators.SuperResolution(3, dim, ratio, 'cuda')
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    loss_a, g = op.data_term(cur, y, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b)
    assert float((loss_a - loss_b).abs().max() / loss_a.abs().max()) < 1e-12
    # persistent-buffer form: only channels [0, C) are written
    buf = torch.full_like(e, 7.0)
    _, _, ge_c = op.fused_last_vjp(xt, e, at, atn, y, g_e_out=buf)
    assert ge_c is buf and torch.equal(buf[:, :3], ge_a[:, :3]) and bool((buf[:, 3:] == 7.0).all())


def test_sr_ratio_32_keeps_the_two_kernel_path():
    from nhmc import operators
    assert not hasattr(operators.SuperResolution(3, 64, 32, 'cuda'), 'fused_last_vjp')
    assert hasattr(operators.SuperResolution(3, 64, 4, 'cuda'), 'fused_last_vjp')


def test_engine_sr_with_and_without_fusion_agree(tiny_score):
    from nhmc import operators, plugin, sampler
    dim, B = 32, 3
    g_ = torch.Generator().manual_seed(11)
    op = operators.SuperResolution(3, dim, 4, 'cuda')
    algo = plugin.HMC(copy.deepcopy(tiny_score).cuda(), op, 0.1)
    eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, torch.device('cuda'))
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    a = eng.decode_and_grad(x, y)
    eng.fuse_last = False
    b = eng.decode_and_grad(x, y)
    for u, v in zip(a, b):
        assert torch.equal(u, v) or float((u - v).abs().max()) <= 1e-12 * float(v.abs().max())


@pytest.mark.parametrize('dim,B', [(32, 2), (64, 3), (256, 2)])
def test_fused_spectral_last_vjp_equals_two_kernel_path(dim, B):
    """The VJP applied in the last product's epilogue = data term (apply_clip = 0 on the clipped decode) followed by
    k_mix_bwd(final_clip = 1): the same MFMA accumulators go through the same scalar op order, hence the same bits."""
    import nhmc.kernels as K
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim)
    op = operators.build_operator('deblur_aniso', 3, dim, torch.device('cuda'))
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    loss_a, g = op.data_term(cur, y, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y, xt_next=cur)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b)
    assert torch.equal(loss_a, loss_b)
    loss_c, gx_c, _ = op.fused_last_vjp(xt, e, at, atn, y)            # decode recomputed inside
    assert torch.equal(gx_a, gx_c) and torch.equal(loss_a, loss_c)
    buf = torch.full_like(e, 7.0)
    op.fused_last_vjp(xt, e, at, atn, y, g_e_out=buf, xt_next=cur)
    assert torch.equal(buf[:, :3], ge_a[:, :3]) and bool((buf[:, 3:] == 7.0).all())


@pytest.mark.parametrize('deg', ['deblur_aniso', 'deblur_gauss'])
def test_engine_spectral_with_and_without_fusion_agree(tiny_score, deg):
    from nhmc import operators, plugin, sampler
    dim, B = 32, 3
    g_ = torch.Generator().manual_seed(13)
    op = operators.build_operator(deg, 3, dim, torch.device('cuda'))
    algo = plugin.HMC(copy.deepcopy(tiny_score).cuda(), op, 0.1)
    eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, torch.device('cuda'))
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    a = eng.decode_and_grad(x, y)
    eng.fuse_last = False
    b = eng.decode_and_grad(x, y)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize('dim,B,factor', [(64, 2, 2), (128, 2, 4), (256, 2, 4)])
def test_fused_srconv_last_vjp_equals_two_kernel_path(dim, B, factor):
    import nhmc.kernels as K
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim + factor)
    op = operators.build_operator(f'sr_bicubic{factor}', 3, dim, torch.device('cuda'))
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    loss_a, g = op.data_term(cur, y, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y, xt_next=cur)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b) and torch.equal(loss_a, loss_b)


@pytest.mark.parametrize('dim,B', [(16, 1), (34, 3), (256, 2)])
def test_fused_color_last_vjp_equals_two_kernel_path(dim, B):
    import nhmc.kernels as K
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim)
    op = operators.Colorization(dim, 'cuda')
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, dim * dim, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    loss_a, g = op.data_term(cur, y, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b)
    assert float((loss_a - loss_b).abs().max() / loss_a.abs().max()) < 1e-12


@pytest.mark.parametrize('dim,B,ratio', [(32, 2, 2), (64, 3, 4), (256, 2, 4)])
def test_fused_cs_last_vjp_equals_two_kernel_path(dim, B, ratio):
    import nhmc.kernels as K
    from nhmc import operators
    g_ = torch.Generator().manual_seed(dim + ratio)
    op = operators.build_operator(f'cs{ratio}', 3, dim, torch.device('cuda'), generator=g_)
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    loss_a, g = op.data_term(cur, y, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y, xt_next=cur)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b) and torch.equal(loss_a, loss_b)


def test_fused_inpaint_pixel_mask_form_equals_slot_form_and_ragged_masks_fall_back():
    import nhmc.kernels as K
    from nhmc import operators
    dim, B = 64, 3
    g_ = torch.Generator().manual_seed(5)
    op = operators.Inpainting(3, dim, oops.random_inpaint_missing(dim, generator=g_), 'cuda')
    assert op.mask_words is not None                                   # whole-pixel mask -> bit mask + prefix counts
    xt = (torch.randn(B, 3, dim, dim, generator=g_) * 0.5).cuda()
    e = torch.randn(B, 6, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    b = osched.betas_fp32()
    at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
    atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
    a = K.ddim_mix_bwd_inpaint(xt, e, at, atn, y, op.slot)
    p = K.ddim_mix_bwd_inpaint_px(xt, e, at, atn, y, op.mask_words, op.mask_prefix)
    for u, v in zip(a, p):
        assert torch.equal(u, v)
    # a mask that drops single channels is not pixel-aligned: the operator keeps the dense slot map
    ragged = operators.Inpainting(3, dim, torch.tensor([0, 5, 7, 100, 3 * dim * dim - 1]), 'cuda')
    assert ragged.mask_words is None
    y2 = torch.randn(B, ragged.M, generator=g_).cuda()
    cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
    la, g = ragged.data_term(cur, y2, apply_clip=False)
    gx_a, ge_a = K.ddim_mix_bwd(g, xt, e, at, atn, final_clip=True)
    lb, gx_b, ge_b = ragged.fused_last_vjp(xt, e, at, atn, y2)
    assert torch.equal(gx_a, gx_b) and torch.equal(ge_a, ge_b)


def test_mid_steps_skip_the_loss_summation_and_change_nothing_else(tiny_score):
    """The sampler reads the per-chain loss at a trajectory's two ends only (main_sampling.py:697,717): a MID step passes
    K.NO_LOSS, the second pass over the data term's tile partials is not launched, and gradients / decode are the same
    bits.  (Summing inside the data-term kernel instead -- last-arriving block per chain, sc1 hand-off -- was measured at
    129 us against 38 us for the two-pass form at 64 chains: DESIGN.md section 3.)"""
    import nhmc.kernels as K
    from nhmc import operators, plugin, sampler
    dim, B = 32, 3
    g_ = torch.Generator().manual_seed(19)
    for deg in ('inpaint_random', 'sr4', 'deblur_aniso', 'color', 'cs4', 'sr_bicubic2'):
        d = 64 if deg == 'sr_bicubic2' else dim
        op = operators.build_operator(deg, 3, d, torch.device('cuda'), generator=g_)
        xt = (torch.randn(B, 3, d, d, generator=g_) * 0.5).cuda()
        e = torch.randn(B, 6, d, d, generator=g_).cuda()
        y = torch.randn(B, op.M, generator=g_).cuda()
        b = osched.betas_fp32()
        at = osched.alpha_bar(b, torch.full((B,), 250)).cuda()
        atn = osched.alpha_bar(b, torch.full((B,), -1)).cuda()
        cur = K.ddim_mix_fwd(xt, e, at, atn, final_clip=True)['xt_next']
        extra = dict(xt_next=cur) if getattr(op, 'fused_wants_decode', False) else {}
        loss_a, gx_a, ge_a = op.fused_last_vjp(xt, e, at, atn, y, **extra)
        loss_b, gx_b, ge_b = op.fused_last_vjp(xt, e, at, atn, y, loss_out=K.NO_LOSS, **extra)
        assert loss_b is None and loss_a is not None
        assert torch.equal(gx_a, gx_b) and torch.equal(ge_a[:, :3], ge_b[:, :3]), deg
        loss_c, g_c = op.data_term(cur, y, apply_clip=False, loss_out=K.NO_LOSS)
        assert loss_c is None
    # through the engine: a trajectory is the same bits, and only FIRST / LAST steps (and the priming) sum a loss
    op = operators.build_operator('inpaint_random', 3, dim, torch.device('cuda'), generator=g_)
    algo = plugin.HMC(copy.deepcopy(tiny_score).cuda(), op, 0.1)
    eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, torch.device('cuda'))
    x = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    p = torch.randn(B, 3, dim, dim, generator=g_).cuda()
    y = torch.randn(B, op.M, generator=g_).cuda()
    eps = torch.full((B,), 0.05, dtype=torch.float64, device='cuda')
    sig = torch.full((B,), 0.9, dtype=torch.float64, device='cuda')
    ws = K.leapfrog_ws(B, x[0].numel(), 'cuda')
    xa, pa, xb, pb = x.clone(), p.clone(), x.clone(), p.clone()
    _, loss_mid = eng.step(K.LF_MID, xa, xa, pa, y, eps, sig, 1.0, ws)
    stale = loss_mid.clone()
    _, loss_full = eng.step(K.LF_MID, xb, xb, pb, y, eps, sig, 1.0, ws, want_loss=True)
    assert torch.equal(xa, xb) and torch.equal(pa, pb)
    _, loss_again = eng.step(K.LF_MID, xa, xa, pa, y, eps, sig, 1.0, ws)
    assert torch.equal(loss_again, loss_full) and stale is not None           # untouched by a MID step: still the last summed loss
