"""Noise schedule on the host (row a8 of SURVEY.md section 8).

The reference recomputes a 1001-element fp32 cumulative product on the device six times per decode
(`compute_alpha`, main_sampling.py:70-73, called from :905-906).  The table depends only on the
betas, so it is built once (sequential fp32 product on the CPU, bit-identical to the reference's
CPU result) and cached per (betas, device); `compute_alpha` keeps the reference's signature.
"""
import numpy as np
import torch


def get_beta_schedule(beta_schedule='linear', *, beta_start=1e-4, beta_end=2e-2, num_diffusion_timesteps=1000):
    """float64 betas (main_sampling.py:36-67; the schedules a config of this path can name)."""
    T = num_diffusion_timesteps
    if beta_schedule == 'linear':
        betas = np.linspace(beta_start, beta_end, T, dtype=np.float64)
    elif beta_schedule == 'quad':
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    elif beta_schedule == 'const':
        betas = beta_end * np.ones(T, dtype=np.float64)
    elif beta_schedule == 'jsd':
        betas = 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    elif beta_schedule == 'sigmoid':
        z = np.linspace(-6, 6, T)
        betas = 1 / (np.exp(-z) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (T,)
    return betas


_tables = {}


def alpha_bar_table(b):
    """fp32 table on b's device, entry k = alpha-bar at t = k-1 (entry 0 = 1).  Cached by the betas' VALUES (the
    host copy is needed for the product anyway), so a recycled device address can never return a stale table."""
    host = b.detach().float().cpu()
    key = (str(b.device), host.numpy().tobytes())
    hit = _tables.get(key)
    if hit is None:
        table = (1 - torch.cat([torch.zeros(1), host])).cumprod(dim=0)
        hit = table.to(b.device)
        if len(_tables) > 16:
            _tables.clear()
        _tables[key] = hit
    return hit


def compute_alpha(beta, t):
    """main_sampling.py:70-73: alpha-bar at LongTensor timesteps t -> [n,1,1,1]; t = -1 gives 1."""
    return alpha_bar_table(beta).index_select(0, t + 1).view(-1, 1, 1, 1)


def timestep_ladder(num_timesteps=1000, timesteps=3):
    """main_sampling.py:469-471."""
    skip = num_timesteps // (timesteps + 1)
    seq = list(range(skip, num_timesteps, skip))
    return seq, [-1] + seq[:-1]
