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


_tables = {}          # id(betas tensor) -> (weakref to it, its version counter, table on its device)
_builds = 0


def table_builds():
    """How many times a table was built (tests: the plugin-surface decode must not rebuild / re-sync per call)."""
    return _builds


def alpha_bar_table(b):
    """fp32 table on b's device, entry k = alpha-bar at t = k-1 (entry 0 = 1): the sequential fp32 product on the CPU,
    bit-identical to the reference's CPU result.  Building it copies the betas to the host (a sync), so it is cached per
    LIVE tensor object and version counter: the same betas tensor handed in again (six times per decode by the
    plugin-surface `iterative_sampling`) costs a dictionary lookup, an in-place edit or a new tensor rebuilds, and a
    recycled address can never return a stale table because the weak reference dies with the tensor."""
    global _builds
    hit = _tables.get(id(b))
    if hit is not None and hit[0]() is b and hit[1] == b._version:
        return hit[2]
    import weakref
    host = b.detach().float().cpu()
    table = (1 - torch.cat([torch.zeros(1), host])).cumprod(dim=0).to(b.device)
    _builds += 1
    for k in [k for k, v in _tables.items() if v[0]() is None]:
        del _tables[k]
    _tables[id(b)] = (weakref.ref(b), b._version, table)
    return table


def compute_alpha(beta, t):
    """main_sampling.py:70-73: alpha-bar at LongTensor timesteps t -> [n,1,1,1]; t = -1 gives 1."""
    return alpha_bar_table(beta).index_select(0, t + 1).view(-1, 1, 1, 1)


def timestep_ladder(num_timesteps=1000, timesteps=3):
    """main_sampling.py:469-471."""
    skip = num_timesteps // (timesteps + 1)
    seq = list(range(skip, num_timesteps, skip))
    return seq, [-1] + seq[:-1]


def mass_tables(n_elem, device, k=1):
    """(std_table, inv_table) by rank r for the diagonal-mass sampler (nhmc_mass_from_variance): sqrt(M_r) and 1 / M_r
    of M_r = exp(k (2 r/(N-1) - 1)), evaluated once on the host with the reference's own tensor expressions
    (main_sampling.py:863-868) -- host constants of N, like the alpha-bar table."""
    ranks = torch.arange(n_elem, dtype=torch.float)
    scores = 2.0 * (ranks / (n_elem - 1)) - 1.0
    M = torch.exp(k * scores)
    return torch.sqrt(M).to(device), (1.0 / M).to(device)
