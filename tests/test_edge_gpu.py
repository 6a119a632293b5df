"""GPU: edge cases of the sampler path -- L = 1 trajectories (FIRST then LAST, no MID), frozen chains (eps_eff = 0),
element counts that are not a multiple of the tile, odd batch sizes against the score chunking."""
import copy

import numpy as np
import pytest
import torch

from oracle import hmc_ref, operators as oops, schedule as osched

pytestmark = pytest.mark.gpu
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _setup(dim, B, tiny_score, chunk=None):
    from nhmc import operators, plugin, sampler
    g_ = torch.Generator().manual_seed(dim * 7 + B)
    missing = oops.random_inpaint_missing(dim, generator=g_)
    ref, op = oops.InpaintRef(3, dim, missing), operators.Inpainting(3, dim, missing, 'cuda')
    algo = plugin.HMC(copy.deepcopy(tiny_score).cuda(), op, 0.1)
    eng = sampler.LeapfrogEngine(algo.score, op, osched.betas_fp32().cuda(), SEQ, SEQ_NEXT, torch.device('cuda'), chunk=chunk)
    x, p = torch.randn(B, 3, dim, dim, generator=g_), torch.randn(B, 3, dim, dim, generator=g_)
    y = ref.H(torch.rand(B, 3, dim, dim, generator=g_) * 2 - 1) + 0.1 * torch.randn(B, ref.M, generator=g_)
    return ref, op, eng, x, p, y


@pytest.mark.parametrize('dim,B,chunk', [(20, 1, None), (20, 5, 2), (32, 7, 3)])
def test_single_leapfrog_step_trajectory_and_ragged_sizes(tiny_score, dim, B, chunk):
    """L = 1; 3*20*20 = 1200 elements (not a multiple of the 2048-element tile); B not a multiple of the chunk."""
    from nhmc import sampler
    ref, op, eng, x, p, y = _setup(dim, B, tiny_score, chunk)
    want = hmc_ref.trajectory(x, p.clone(), osched.betas_fp32(), SEQ, SEQ_NEXT, tiny_score, ref, y, sigma_y=0.7, eps=0.05, m=1.0, L=1)
    st = sampler.ChainState(B, 1.0, 0.05, 'cuda')
    st['eps_eff'].fill_(0.05)
    st['sigma_y'].fill_(0.7)
    got = sampler.run_trajectory(eng, x.cuda(), p.cuda().clone(), y.cuda(), st, 1.0, 1)
    assert rel(got['x_prop'], want['x']) < 1e-5 and rel(got['p'], want['p']) < 1e-4 and rel(got['xt'], want['xt']) < 1e-5
    for k in ('H0', 'H1'):
        assert float((got[k].cpu() - want[k]).abs().max()) <= 8 * float(np.spacing(np.float32(want[k].abs().max())))


def test_frozen_chains_do_not_move(tiny_score):
    """A finished chain runs with eps_eff = 0: its position is untouched by a whole trajectory and it never accepts."""
    import nhmc.kernels as K
    from nhmc import sampler
    ref, op, eng, x, p, y = _setup(32, 3, tiny_score)
    st = sampler.ChainState(3, 1.0, 0.05, 'cuda')
    st['epoch'].copy_(torch.tensor([0, 100, 5], dtype=torch.int32))
    K.schedule_begin(st, 0.1, 60, 20)
    assert st['active'].cpu().tolist() == [1, 0, 1] and st['eps_eff'].cpu().tolist() == [0.05, 0.0, 0.05]
    dx = x.cuda()
    got = sampler.run_trajectory(eng, dx, p.cuda().clone(), y.cuda(), st, 1.0, 3)
    assert torch.equal(got['x_prop'][1], dx[1]) and not torch.equal(got['x_prop'][0], dx[0])
    acc, _ = K.metropolis(got['H0'], got['H1'], torch.zeros(3, device='cuda'), st['active'])
    assert int(acc[1]) == 0


def test_copy_probe_copies_and_checks_alignment():
    import nhmc.kernels as K
    from nhmc import _lib
    src = torch.randn(3, 1000, device='cuda')                     # 3000 elements: a ragged last block
    dst = torch.zeros_like(src)
    assert torch.equal(K.copy_probe(src, dst), src)
    flat = torch.randn(4001, device='cuda')
    with pytest.raises(_lib.NhmcError):
        K.copy_probe(flat[1:], torch.zeros(4000, device='cuda'))  # misaligned source
    with pytest.raises(_lib.NhmcError):
        K.copy_probe(flat[:3998].contiguous(), torch.zeros(3998, device='cuda'))   # n % 4 != 0
