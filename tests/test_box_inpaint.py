"""Box inpainting (main_sampling.py:290-299): mask construction on the host and the same kernels on the GPU."""
import random

import pytest
import torch

from oracle import hmc_ref, operators as oops


def test_box_mask_matches_reference_construction():
    from nhmc import operators
    random.seed(7)
    op = operators.build_operator('inpaint_box', 3, 256, 'cpu')
    random.seed(7)
    left, up = random.randint(16, 112), random.randint(16, 112)
    missing = torch.zeros(256, 256, 3)
    missing[left:left + 128, up:up + 128, :] = 1.0
    want = torch.nonzero(missing.view(-1)).squeeze()
    assert torch.equal(op.missing_indices, want) and op.M == 3 * (256 * 256 - 128 * 128)


@pytest.mark.gpu
def test_box_inpainting_on_the_gpu():
    from nhmc import operators
    random.seed(3)
    op = operators.build_operator('inpaint_box', 3, 256, 'cuda')
    ref = oops.InpaintRef(3, 256, op.missing_indices.cpu())
    g_ = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 256, 256, generator=g_) * 0.8
    y = torch.randn(2, ref.M, generator=g_)
    assert torch.equal(op.H(x.cuda()).cpu(), ref.H(x)) and torch.equal(op.Ht(y.cuda()).cpu(), ref.Ht(y))
    loss_ref, g_ref = hmc_ref.data_term(x, ref, y)
    loss, g = op.data_term(x.cuda(), y.cuda(), apply_clip=True)
    assert torch.equal(g.cpu(), g_ref) and float((loss.cpu() - loss_ref.double()).abs().max() / loss_ref.max()) < 1e-6
