"""Oracle: noise schedule (test infrastructure, see oracle/__init__.py).

Restates
  * `get_beta_schedule`  main_sampling.py:36-67   (only the schedules the FFHQ
    config can select are kept: linear / quad / const)
  * `compute_alpha`      main_sampling.py:70-73   (alpha-bar lookup: fp32
    cumulative product of 1-[0, beta], gathered at t+1 so t=-1 gives exactly 1)
  * the timestep ladder  main_sampling.py:469-471 (skip = T // timesteps)
"""
import numpy as np
import torch


def beta_schedule(kind="linear", beta_start=1e-4, beta_end=2e-2, steps=1000):
    """float64 numpy betas, as the reference builds them before `.float()`."""
    if kind == "linear":
        return np.linspace(beta_start, beta_end, steps, dtype=np.float64)
    if kind == "quad":
        return np.linspace(beta_start ** 0.5, beta_end ** 0.5, steps, dtype=np.float64) ** 2
    if kind == "const":
        return beta_end * np.ones(steps, dtype=np.float64)
    raise NotImplementedError(kind)


def betas_fp32(**kw):
    """`torch.from_numpy(betas).float()` as handed to `hmc` (main_sampling.py:362,475)."""
    return torch.from_numpy(beta_schedule(**kw)).float()


def alpha_bar(b, t):
    """alpha-bar at integer timesteps `t` (LongTensor [n]) -> [n,1,1,1] fp32.

    main_sampling.py:70-73: a zero is prepended to beta so that index t+1 is the
    product over beta_0..beta_t and t = -1 selects the empty product (1.0).
    """
    padded = torch.cat([torch.zeros(1, dtype=b.dtype), b], dim=0)
    table = (1 - padded).cumprod(dim=0)
    return table.index_select(0, t + 1).view(-1, 1, 1, 1)


def alpha_bar_table(b):
    """The whole fp32 lookup table, entry k = alpha-bar at t = k-1."""
    padded = torch.cat([torch.zeros(1, dtype=b.dtype), b], dim=0)
    return (1 - padded).cumprod(dim=0)


def timestep_ladder(num_timesteps=1000, timesteps=3):
    """main_sampling.py:469-471 -> (seq, seq_next), e.g. [250,500,750] / [-1,250,500]."""
    skip = num_timesteps // (timesteps + 1)
    seq = list(range(skip, num_timesteps, skip))
    return seq, [-1] + seq[:-1]
