"""Oracle: diagonal-mass HMC (test infrastructure, see oracle/__init__.py).

Restates `hmc_test_conditioning` main_sampling.py:776-894 -- an experiment the reference keeps but never dispatches
(`:481-483` call `hmc` only): burn-in 5 + 40 annealing + 4*10 sampling epochs, sigma_y = sigma_0 + 0.9 during burn-in
then a cubic decay (:808-816), a diagonal mass M (momentum ~ N(0, M), kinetic term sum p^2/M, position step eps p/M),
Welford mean/variance of the trajectory positions once (epoch-burn) > epochs//3 (:843-847), and on accept with
epoch > epochs//3 the mass is rebuilt from the rank transform of that variance: M = exp(2 rank/(N-1) - 1) (:857-870).
"""
import math

import torch

from .hmc_ref import _data_loss_and_grad


def mass_from_variance(M2, L, k=1, stable=False, tables=None):
    """main_sampling.py:858-870 for one flattened chain: -> (M, std, inv_M), all [N].
    tables (optional): (M_by_rank, std_by_rank) of the host the reference ran on (fixture G11b) in place of this
    host's torch.exp / torch.sqrt -- M is a function of the rank alone.

    The reference sorts with torch's default UNSTABLE sort (:860), so among equal variances -- e.g. the all-zero
    variance of epochs 14..18, before the Welford accumulation starts -- the rank order is whatever the sort
    implementation of the day produces (CPU and GPU torch disagree).  `stable=True` breaks ties by index, which is
    what the build's radix sort does; the G11 fixture replays the reference run with `stable=False`."""
    variance = (M2 / (L - 1)).view(-1)
    _, sorted_idx = torch.sort(variance, stable=stable)
    ranks = torch.zeros_like(sorted_idx, dtype=torch.float)
    ranks[sorted_idx] = torch.arange(len(variance), dtype=torch.float)
    if tables is not None:
        M, std = tables[0][ranks.long()], tables[1][ranks.long()]
        return M, std, 1.0 / M
    scores = 2.0 * (ranks / (variance.numel() - 1)) - 1.0
    M = torch.exp(k * scores)
    return M, torch.sqrt(M), 1.0 / M


def sigma_y_mass(epoch, sigma_0, burn=5, epochs=40):
    if epoch < burn:
        return sigma_0 + 0.9
    return sigma_0 + 0.9 * (1 - (epoch - burn) / epochs) ** 3


def hmc_mass_reference(x, b, seq, seq_next, model, Hop, y_0, x_orig, *, tau, epsilon, sigma_0, burn=5, epochs=40,
                       sampling=10, trace=None, stable_sort=False, tables=None):
    """Batch-1 loop of main_sampling.py:776-894 with the reference's RNG call order."""
    x = x.detach().requires_grad_()
    L = max(1, math.floor(tau / epsilon))
    finals, rejected, epoch, sigma_y = [], 0, 0, None
    M_diag = torch.ones_like(x).reshape(-1)
    std_diag, inv_M = torch.sqrt(M_diag), 1.0 / M_diag
    while epoch < burn + epochs + 4 * sampling:
        mean, M2 = torch.zeros_like(x), torch.zeros_like(x)
        if epoch < burn:
            sigma_y = sigma_0 + 0.9
        elif epoch < epochs:
            sigma_y = sigma_0 + 0.9 * (1 - (epoch - burn) / epochs) ** 3
        elif epoch == epochs:
            sigma_y = sigma_0
            if tau > 0.1:
                tau = 0.1
                epsilon = 0.01
        p = torch.randn_like(x) * std_diag.view_as(x)
        _, loss, _, g = _data_loss_and_grad(x, b, seq, seq_next, model, Hop, y_0)
        k = 1 / (2 * sigma_y ** 2)
        H0 = (1 / 2) * torch.sum(x.detach() ** 2, dim=(1, 2, 3)) + k * loss + (1 / 2) * torch.sum(inv_M.view_as(x) * p ** 2)
        xp = x.detach().clone().requires_grad_(True)
        p = p - (epsilon / 2) * (xp.detach() + k * g)
        for l in range(L):
            xp = xp + epsilon * p * inv_M.view_as(xp)
            xp = xp.detach().requires_grad_(True)
            xt, loss, _, g = _data_loss_and_grad(xp, b, seq, seq_next, model, Hop, y_0)
            p = p - epsilon * (xp.detach() + k * g)
            if (epoch - burn) > epochs // 3:
                delta = xp.detach() - mean
                mean = mean + delta / (l + 1)
                delta2 = xp.detach() - mean
                M2 = M2 + delta * delta2
        p = p + (epsilon / 2) * (xp.detach() + k * g)
        H1 = (1 / 2) * torch.sum(xp.detach() ** 2, dim=(1, 2, 3)) + k * loss + (1 / 2) * torch.sum(inv_M.view_as(xp) * p ** 2)
        dH = H1 - H0
        ratio = min(torch.tensor([1.0]), torch.exp(-dH))
        accept = torch.rand(1).item() < ratio.item()
        if trace is not None:
            trace.setdefault('dH', []).append(float(dH.item()))
            trace.setdefault('accept', []).append(bool(accept))
            trace.setdefault('epoch', []).append(epoch)
        if accept:
            if epoch > epochs // 3:
                _, std_diag, inv_M = mass_from_variance(M2, L, stable=stable_sort, tables=tables)
            rejected = 0
            if epoch >= epochs + sampling:
                finals.append(xt.clone()[0])
            epoch += 1
            x = xp.detach().clone().requires_grad_(True)
        else:
            rejected += 1
            if rejected >= 2:
                tau = tau * 0.95
                epsilon = epsilon * 0.95
    return torch.stack(finals)
