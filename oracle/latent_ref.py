"""Oracle: latent-space variant (test infrastructure, see oracle/__init__.py).

Restates `hmc_latent` main_sampling_latent.py:623-762 and its `iterative_sampling` :765-789 with the
`Unconditional_Latent` plugin (algos/unconditional_latent.py) inlined.  The model is duck-typed exactly as the
reference uses it: `apply_model(x, t, cond)`, `differentiable_decode_first_stage(z)`, `alphas_cumprod`,
`alphas_cumprod_prev`.

Differences from the pixel sampler that are kept on purpose (they are what the reference does):
  * `for epoch in range(epochs + 2*sampling)`: a rejected trajectory consumes its epoch (:646,733);
  * sigma_y starts at opt.sigma_y and is re-set only on accept, geometrically towards sigma_0 (:693-695),
    then sigma_0 with tau = 0.1, eps = 0.01 (:705-708);
  * from the second consecutive reject: tau, eps *= 0.9 and the counter is reset (:728-732);
  * the sample appended at an accept with epoch >= epochs is the PREVIOUS accepted latent `x_accept[0]`
    (:709, before `x_accept` is reassigned at :713); the last 10 are returned, as latents (:760-762).
"""
import math

import numpy as np
import torch

from .ddim import ddim_step
from .hmc_ref import hamiltonian, _col, to_unit_range, psnr_unit


def alpha_table(model):
    """main_sampling_latent.py:771-773: entry k = alpha-bar at t = k-1."""
    return torch.cat([model.alphas_cumprod_prev[0:1], model.alphas_cumprod], dim=0)


def decode_latent(x, seq, seq_next, model):
    """main_sampling_latent.py:765-789."""
    n = x.shape[0]
    table = alpha_table(model)
    xt = x
    for i, j in zip(reversed(seq), reversed(seq_next)):
        t = torch.ones(n) * i
        at = torch.full((n, 1, 1, 1), float(table[i + 1]))
        at_next = torch.full((n, 1, 1, 1), float(table[j + 1]))
        et = model.apply_model(xt, t, None)
        xt = ddim_step(xt, et, at, at_next)
    return xt


def _loss_and_grad(x_leaf, seq, seq_next, model, Hop, y_0):
    xt = decode_latent(x_leaf, seq, seq_next, model).clip(-1, 1)
    resid = y_0 - Hop.H(model.differentiable_decode_first_stage(xt))
    per_chain = (resid ** 2).reshape(resid.shape[0], -1).sum(dim=1)
    loss = torch.sum(resid ** 2)
    grad = torch.autograd.grad(loss, x_leaf, retain_graph=False)[0]
    return xt.detach(), loss.detach(), per_chain.detach(), grad


def hmc_latent_reference(x, seq, seq_next, model, Hop, y_0, x_orig, *, sigma_y, tau, epsilon, m, sigma_0,
                         epochs=50, sampling=10, trace=None):
    """Batch-1 loop of main_sampling_latent.py:623-762 with the reference's RNG call order."""
    x = x.detach().requires_grad_()
    sigma_y0 = sigma_y
    L = max(1, math.floor(tau / epsilon))
    finals, rejected, x_accept = [], 0, None
    for epoch in range(epochs + 2 * sampling):
        p = torch.randn_like(x) * math.sqrt(m)
        _, loss, _, g = _loss_and_grad(x, seq, seq_next, model, Hop, y_0)
        k = 1 / (2 * sigma_y ** 2)
        H0 = (1 / 2) * torch.sum(x.detach() ** 2, dim=(1, 2, 3)) + k * loss + (1 / 2) * torch.sum(p * p, dim=(1, 2, 3)) * m ** (-1)
        xp = x.detach().clone().requires_grad_(True)
        p = p - (epsilon / 2) * (xp.detach() + k * g)
        for _ in range(L):
            xp = xp + epsilon * m ** (-1) * p
            xp = xp.detach().requires_grad_(True)
            xt, loss, _, g = _loss_and_grad(xp, seq, seq_next, model, Hop, y_0)
            p = p - epsilon * (xp.detach() + k * g)
        p = p + (epsilon / 2) * (xp.detach() + k * g)
        H1 = (1 / 2) * torch.sum(xp.detach() ** 2, dim=(1, 2, 3)) + k * loss + (1 / 2) * torch.sum(p * p, dim=(1, 2, 3)) * m ** (-1)
        dH = H1 - H0
        ratio = min(torch.tensor([1.0]), torch.exp(-dH))
        accept = torch.rand(1).item() < ratio.item()
        if trace is not None:
            trace.setdefault('dH', []).append(float(dH.item()))
            trace.setdefault('accept', []).append(bool(accept))
            trace.setdefault('sigma_y', []).append(float(sigma_y))
            trace.setdefault('eps', []).append(float(epsilon))
        if accept:
            rejected = 0
            if epoch < epochs:
                sigma_y = sigma_y0 * (sigma_0 / sigma_y0) ** (epoch / epochs)
            else:
                sigma_y = sigma_0
                tau = 0.1
                epsilon = 0.01
                finals.append(x_accept[0])
            x_accept = xt.clone()
            x = xp.detach().clone().requires_grad_(True)
        else:
            rejected += 1
            if rejected >= 2:
                tau = tau * 0.9
                epsilon = epsilon * 0.9
                rejected = 0
    return torch.stack(finals[-sampling:])


def trajectory_latent(x, p, seq, seq_next, model, Hop, y_0, *, sigma_y, eps, m, L):
    """One outer iteration, per-chain loss / H (the form the GPU engine is compared with)."""
    B = x.shape[0]
    sig = np.broadcast_to(np.asarray(sigma_y, dtype=np.float64), (B,))
    ep = np.broadcast_to(np.asarray(eps, dtype=np.float64), (B,))
    k64 = 1 / (2 * sig ** 2)
    kf, eh, ef, ex = _col(k64, B), _col(ep / 2, B), _col(ep, B), _col(ep * m ** (-1), B)
    x0 = x.detach().clone().requires_grad_(True)
    _, _, loss_b, g = _loss_and_grad(x0, seq, seq_next, model, Hop, y_0)
    H0 = hamiltonian(torch.sum(x0.detach() ** 2, dim=(1, 2, 3)), loss_b, torch.sum(p * p, dim=(1, 2, 3)), k64, m)
    xp = x0.detach().clone()
    p = p - eh * (xp + kf * g)
    xt = None
    for _ in range(L):
        xp = (xp + ex * p).detach().requires_grad_(True)
        xt, _, loss_b, g = _loss_and_grad(xp, seq, seq_next, model, Hop, y_0)
        xp = xp.detach()
        p = p - ef * (xp + kf * g)
    p = p + eh * (xp + kf * g)
    H1 = hamiltonian(torch.sum(xp ** 2, dim=(1, 2, 3)), loss_b, torch.sum(p * p, dim=(1, 2, 3)), k64, m)
    return dict(x=xp, p=p, xt=xt, loss=loss_b, H0=H0, H1=H1)


class TinyLatentModel(torch.nn.Module):
    """Duck-typed stand-in for the LatentDiffusion object (its real class needs pytorch_lightning + taming,
    absent offline): a tiny score network on the latent and a tiny x4 decoder, seeded weights."""

    def __init__(self, seed=77, steps=1000):
        super().__init__()
        from .tiny_score import TinyScore
        g = torch.Generator().manual_seed(seed)
        self.score = TinyScore(out_ch=3)
        self.up1 = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.up2 = torch.nn.Conv2d(8, 3, 3, padding=1)
        with torch.no_grad():
            for prm in self.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.35 / math.sqrt(max(1, prm[0].numel()))))
        betas = torch.linspace(0.0015 ** 0.5, 0.0195 ** 0.5, steps, dtype=torch.float64) ** 2      # LDM "linear" schedule
        ac = torch.cumprod(1 - betas, dim=0)
        self.register_buffer('alphas_cumprod', ac.float())
        self.register_buffer('alphas_cumprod_prev', torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]]).float())
        self.eval().requires_grad_(False)

    def apply_model(self, x, t, cond=None):
        return self.score(x, t)

    def differentiable_decode_first_stage(self, z):
        h = torch.nn.functional.interpolate(z, scale_factor=4, mode='nearest')
        return torch.tanh(self.up2(torch.tanh(self.up1(h))))
