"""CPU: the diagonal-mass oracle (oracle/mass_ref.py) against the reference's own `hmc_test_conditioning` run (G11)."""
import numpy as np
import torch

from oracle import mass_ref, operators as oops, schedule

T = torch.from_numpy


def test_g11_full_diagonal_mass_run_bit_exact(golden, tiny_score):
    g = golden('g11_hmc_mass_16.npz')
    op = oops.InpaintRef(3, 16, T(g['missing']))
    torch.manual_seed(int(g['seed']))
    trace = {}
    out = mass_ref.hmc_mass_reference(T(g['x']), schedule.betas_fp32(), [250, 500, 750], [-1, 250, 500], tiny_score, op,
                                      T(g['y_0']), T(g['x_orig']), tau=float(g['tau']), epsilon=float(g['epsilon']),
                                      sigma_0=float(g['sigma_0']), trace=trace)
    assert out.shape == g['out'].shape == (35, 3, 16, 16)
    assert np.array_equal(out.numpy(), g['out'])
    assert np.array_equal(-np.array(trace['dH'], dtype=np.float32), g['neg_dH'].astype(np.float32))


def test_rank_transform_of_the_variance():
    M2 = torch.tensor([[3.0, 1.0, 2.0, 5.0, 4.0]])
    M, std, inv = mass_ref.mass_from_variance(M2, L=3)
    ranks = torch.tensor([2.0, 0.0, 1.0, 4.0, 3.0])
    assert torch.allclose(M, torch.exp(2 * ranks / 4 - 1)) and torch.allclose(std ** 2, M) and torch.allclose(inv * M, torch.ones(5))
