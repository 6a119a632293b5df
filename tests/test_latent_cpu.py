"""CPU: the latent-space oracle (oracle/latent_ref.py) against the reference's own `hmc_latent` run (G7)."""
import numpy as np
import torch

from oracle import latent_ref, operators as oops

T = torch.from_numpy
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def test_g7_full_hmc_latent_bit_exact(golden):
    g = golden('g7_hmc_latent_16.npz')
    model = latent_ref.TinyLatentModel()
    op = oops.InpaintRef(3, 64, T(g['missing']))
    torch.manual_seed(int(g['seed']))
    trace = {}
    out = latent_ref.hmc_latent_reference(T(g['x']), SEQ, SEQ_NEXT, model, op, T(g['y_0']), T(g['x_orig']),
                                          sigma_y=float(g['sigma_y']), tau=float(g['tau']), epsilon=float(g['epsilon']),
                                          m=float(g['m']), sigma_0=float(g['sigma_0']), trace=trace)
    assert out.shape == g['out'].shape and np.array_equal(out.numpy(), g['out'])
    assert np.array_equal(-np.array(trace['dH'], dtype=np.float32), g['neg_dH'].astype(np.float32))
    assert len(trace['accept']) == 70


def test_g7_first_trajectory_per_chain_form(golden):
    g = golden('g7_hmc_latent_16.npz')
    model = latent_ref.TinyLatentModel()
    op = oops.InpaintRef(3, 64, T(g['missing']))
    out = latent_ref.trajectory_latent(T(g['x']), T(g['p0']), SEQ, SEQ_NEXT, model, op, T(g['y_0']),
                                       sigma_y=float(g['sigma_y']), eps=float(g['epsilon']), m=1.0,
                                       L=int(float(g['tau']) / float(g['epsilon'])))   # floor(0.3/0.1) == 2 in fp64
    # per-chain loss is a row-wise sum (the reference sums the whole batch): same value to fp32 rounding of H
    assert abs(float((out['H1'] - out['H0'])[0]) + float(g['neg_dH'][0])) < 0.01
