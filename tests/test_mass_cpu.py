"""CPU: the diagonal-mass oracle (oracle/mass_ref.py) against the reference's own `hmc_test_conditioning` run (G11)."""
import numpy as np
import pytest
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


@pytest.mark.parametrize('fixture,dim', [('g11b_hmc_mass_stable_16.npz', 16), ('g11c_hmc_mass_stable_64.npz', 64)])
def test_g11b_reference_run_with_ties_broken_by_index_bit_exact(golden, tiny_score, fixture, dim):
    """G11b (oracle/gen_golden_mass.py stable): the reference's `hmc_test_conditioning` run with torch.sort made stable by
    the generator's wrapper and the float64 tiny score -- the tie rule the product's radix sort implements.  The oracle's
    stable-sort mode must be that run bit for bit; it is what tests/test_mass_gpu.py compares the GPU loop with."""
    from oracle.tiny_score import F64Score
    g = golden(fixture)                                                  # G11c: the same run at 64 x 64 (12 288 elements per rank transform)
    assert int(g['sorts']) > 10 and int(g['tied_elements']) > 0          # rank transforms happened, some among tied variances
    N = g['x'].size
    ranks = torch.arange(N, dtype=torch.float)
    M_here = torch.exp(1 * (2.0 * (ranks / (N - 1)) - 1.0))
    same_libm = np.array_equal(M_here.numpy(), g['M_by_rank']) and np.array_equal(torch.sqrt(M_here).numpy(), g['std_by_rank'])
    if not same_libm:
        print('this host\'s torch.exp / torch.sqrt differ from the generating host\'s: using the recorded mass tables')
    op = oops.InpaintRef(3, dim, T(g['missing']))
    torch.manual_seed(int(g['seed']))
    trace = {}
    out = mass_ref.hmc_mass_reference(T(g['x']), schedule.betas_fp32(), [250, 500, 750], [-1, 250, 500], F64Score(tiny_score), op,
                                      T(g['y_0']), T(g['x_orig']), tau=float(g['tau']), epsilon=float(g['epsilon']),
                                      sigma_0=float(g['sigma_0']), trace=trace, stable_sort=True,
                                      tables=None if same_libm else (T(g['M_by_rank']), T(g['std_by_rank'])))
    assert len(trace['accept']) == len(g['u'])
    assert np.array_equal(out.numpy(), g['out'])
    assert np.array_equal(-np.array(trace['dH'], dtype=np.float32), g['neg_dH'].astype(np.float32))
    # and it is a DIFFERENT realisation from the default-sort run (G11): the tie order matters
    if dim == 16:
        assert not np.array_equal(g['out'], golden('g11_hmc_mass_16.npz')['out'])


def test_rank_transform_of_the_variance():
    M2 = torch.tensor([[3.0, 1.0, 2.0, 5.0, 4.0]])
    M, std, inv = mass_ref.mass_from_variance(M2, L=3)
    ranks = torch.tensor([2.0, 0.0, 1.0, 4.0, 3.0])
    assert torch.allclose(M, torch.exp(2 * ranks / 4 - 1)) and torch.allclose(std ** 2, M) and torch.allclose(inv * M, torch.ones(5))
