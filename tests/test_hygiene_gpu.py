"""GPU: coverage the round-1 review asked for -- the kernels' square roots over the whole alpha-bar table, the DDIM mix
bit-exact on the CLI's default 10-step ladder, the score network (forward and input gradient, eager and hipGraph) on the
MI355X against the reference-class fixture G6, and the alpha-bar table cache of the plugin surface."""
import numpy as np
import pytest
import torch

from oracle import ddim, schedule
from tests.test_kernels_gpu import dev, gen, rel, same_bits

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_kernel_square_roots_are_correctly_rounded_over_the_whole_table():
    """sqrt(alpha-bar) and sqrt(1 - alpha-bar) for all 1001 table entries, read back through the kernels
    (map_back with x0 = 1, add = 0 gives sqrt(at_next); add_up with e = 1 gives sqrt(1 - at_next)), against numpy's
    correctly rounded fp32 sqrt.  This is what makes the elementwise kernels bit-exact for ANY timestep ladder."""
    import nhmc.kernels as K
    table = schedule.alpha_bar_table(schedule.betas_fp32())                       # [1001]
    B = table.numel()
    ones, zeros = torch.ones(B, 1, 2, 2), torch.zeros(B, 1, 2, 2)
    c3 = K.ddim_map_back(dev(ones), dev(zeros), table)[:, 0, 0, 0].cpu().numpy()
    assert np.array_equal(c3, np.sqrt(table.numpy()))
    out = K.ddim_mix_fwd(dev(zeros), dev(ones), table.clamp_min(1e-3), table, want=('add_up',))['add_up'][:, 0, 0, 0].cpu().numpy()
    assert np.array_equal(out, np.sqrt((1 - table).numpy()))


def test_ddim_mix_bit_exact_on_the_default_ten_step_ladder():
    """`--timesteps 10` (the CLI default, main_sampling.py:975): every (t, t_next) pair of that ladder, one per chain."""
    import nhmc.kernels as K
    seq, seq_next = schedule.timestep_ladder(1000, 10)
    b = schedule.betas_fp32()
    at, atn = schedule.alpha_bar(b, torch.tensor(seq)), schedule.alpha_bar(b, torch.tensor(seq_next))
    B, g_ = len(seq), gen(31)
    shape = (B, 3, 32, 32)
    xt = torch.randn(shape, generator=g_).requires_grad_(True)
    e = torch.randn(B, 6, 32, 32, generator=g_).requires_grad_(True)
    gout = torch.randn(shape, generator=g_)
    x0, add = ddim.predict_x0(xt.detach(), e.detach(), at, atn)
    out = K.ddim_mix_fwd(dev(xt.detach()), dev(e.detach()), at, atn, want=('xt_next', 'x0_t', 'add_up'))
    same_bits(out['x0_t'], x0, 'x0_t')
    same_bits(out['add_up'], add, 'add_up')
    same_bits(out['xt_next'], ddim.renoise(x0, add, atn), 'xt_next')
    for final_clip in (False, True):
        nxt = ddim.ddim_step(xt, e, at, atn)
        nxt = nxt.clip(-1, 1) if final_clip else nxt
        ga, gb = torch.autograd.grad(nxt, (xt, e), gout)
        dx, de = K.ddim_mix_bwd(dev(gout), dev(xt.detach()), dev(e.detach()), at, atn, final_clip=final_clip)
        same_bits(dx, ga, 'g_xt')
        same_bits(de, gb, 'g_e')
    # score without gradient: the direct piece alone, identical bits, no g_e
    dx2, none = K.ddim_mix_bwd(dev(gout), dev(xt.detach()), dev(e.detach()), at, atn, want_g_e=False)
    assert none is None
    same_bits(dx2, torch.autograd.grad(ddim.ddim_step(xt, e, at, atn), xt, gout)[0], 'g_xt without g_e')


def test_score_network_on_the_gpu_matches_the_reference_unet(golden):
    """G6 on the MI355X: forward and input gradient of nhmc.unet (MIOpen-backed) against the reference's own U-Net class
    (CPU fixture); the engine's decode + gradient with it eager and replayed as a hipGraph."""
    from nhmc import operators, sampler, unet
    g = golden('g6_unet_64.npz')
    cfg = dict(image_size=64, num_channels=32, num_res_blocks=1, channel_mult='', learn_sigma=True,
               attention_resolutions='16', num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True)
    net = unet.create_model(**cfg).eval()
    wg = torch.Generator().manual_seed(int(g['weight_seed']))
    net.load_state_dict({k: torch.randn(v.shape, generator=wg) * float(g['weight_scale']) for k, v in net.state_dict().items()})
    net = net.cuda().requires_grad_(False)
    x = T(g['x']).cuda().requires_grad_(True)
    out = net(x, T(g['t']).cuda())
    (gx,) = torch.autograd.grad(out, x, T(g['gout']).cuda())
    assert rel(out, T(g['out'])) < 1e-4 and rel(gx, T(g['gx'])) < 1e-4
    # the engine's decode + gradient with this network: eager vs hipGraph replay
    op = operators.SuperResolution(3, 64, 4, 'cuda')
    b = schedule.betas_fp32().cuda()
    eng = sampler.LeapfrogEngine(net, op, b, [250, 500, 750], [-1, 250, 500], torch.device('cuda'))
    y = torch.randn(2, op.M, generator=gen(2)).cuda()
    xs = T(g['x']).cuda()
    eager = eng.decode_and_grad(xs, y)
    graphed = eng.decode_and_grad(xs, y, graph=True)
    for a, b_ in zip(eager, graphed):                 # same kernels; MIOpen may pick another solver inside the capture
        assert rel(b_, a) < 1e-5


def test_alpha_table_cache_does_not_resync_the_host():
    """`compute_alpha` keeps the reference's signature (main_sampling.py:70-73) but must not copy the betas to the host
    on every call: the table is cached per live tensor + version, and an in-place edit invalidates it."""
    from nhmc import schedule as S
    b = schedule.betas_fp32().cuda()
    t = torch.tensor([749, 499, 249, -1], device='cuda')
    want = schedule.alpha_bar(schedule.betas_fp32(), t.cpu())
    first = S.compute_alpha(b, t)
    builds = S.table_builds()
    for _ in range(5):
        again = S.compute_alpha(b, t)
    assert S.table_builds() == builds and torch.equal(first.cpu(), want) and torch.equal(again.cpu(), want)
    b.mul_(0.5)                                                   # in-place edit: version bump -> rebuilt
    half = S.compute_alpha(b, t)
    assert S.table_builds() == builds + 1
    assert torch.equal(half.cpu(), schedule.alpha_bar(schedule.betas_fp32() * 0.5, t.cpu()))
