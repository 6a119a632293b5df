"""Oracle: the noise-space HMC sampler (test infrastructure, see oracle/__init__.py).

Restates `hmc` main_sampling.py:660-774 three ways:

  * `hmc_reference`  -- the reference's batch-1 loop with the reference's RNG call
    order (device `randn_like` for the momentum, then CPU `torch.rand(1)` for the
    accept test), so that under the same torch seed it reproduces the reference's
    returned tensor bit for bit on CPU;
  * `trajectory`     -- one outer iteration (momentum half step, L leapfrog steps,
    both Hamiltonians) with PER-CHAIN loss / H, the momentum given as an input;
  * `hmc_chains`     -- the per-chain generalisation of the whole loop (every chain
    its own epoch counter, sigma_y, epsilon, tau and reject counter, all chains
    advancing one trajectory per iteration) with the noise supplied by the caller.
    At B = 1 it takes exactly the decisions `hmc_reference` takes.

Scalars follow the reference: tau, epsilon, sigma_y, m are Python floats (fp64)
and every `scalar * tensor` is an fp32 multiply by the scalar rounded to fp32.
"""
import math

import numpy as np
import torch

from .ddim import decode


def to_unit_range(x):
    """datasets/__init__.py:214-223 with `rescaled: true`: clamp((x+1)/2, 0, 1)."""
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)


def psnr_unit(xhat01, ref01):
    """main_sampling.py:738-739."""
    mse = torch.mean((xhat01 - ref01) ** 2)
    return 10 * torch.log10(1 / mse)


def sigma_y_at(epoch, sigma_0, epochs=60):
    """main_sampling.py:683-685 (annealing phase only)."""
    return sigma_0 + 1.6 * (1 - epoch / epochs) ** 2


def _data_loss_and_grad(x_leaf, b, seq, seq_next, model, Hop, y_0):
    """main_sampling.py:693-695 / :709-711: decode, final clip, sum of squared
    residuals over the WHOLE batch, gradient w.r.t. the noise."""
    xt = decode(x_leaf, b, seq, seq_next, model).clip(-1, 1)
    resid = y_0 - Hop.H(xt)
    per_chain = (resid ** 2).reshape(resid.shape[0], -1).sum(dim=1)
    loss = torch.sum(resid ** 2)
    grad = torch.autograd.grad(loss, x_leaf, retain_graph=False)[0]
    return xt.detach(), loss.detach(), per_chain.detach(), grad


def hmc_reference(x, b, seq, seq_next, model, Hop, y_0, x_orig, *, tau, epsilon, m, sigma_0,
                  epochs=60, sampling=20, trace=None):
    """Batch-1 reference loop, main_sampling.py:660-774.  `sigma_0` is the value
    AFTER the reference's doubling (main_sampling.py:348).

    trace (optional dict) receives: 'psnr', 'dH', 'accept', 'sigma_y', 'eps'."""
    x = x.detach().requires_grad_()
    L = max(1, math.floor(tau / epsilon))                       # :664, never recomputed
    orig01 = [to_unit_range(x_orig[j]) for j in range(len(x_orig))]
    finals = []
    rejected = 0
    epoch = 0
    sigma_y = None
    while epoch < epochs + 2 * sampling:                        # :681
        if epoch < epochs:                                      # :683-689
            sigma_y = sigma_0 + 1.6 * (1 - epoch / epochs) ** 2
        elif epoch == epochs:
            sigma_y = sigma_0
            if tau > 0.1:
                tau = 0.1
                epsilon = 0.01

        p = torch.randn_like(x) * math.sqrt(m)                  # :692
        _, loss, _, g = _data_loss_and_grad(x, b, seq, seq_next, model, Hop, y_0)
        k = 1 / (2 * sigma_y ** 2)
        H0 = (1 / 2) * torch.sum(x.detach() ** 2, dim=(1, 2, 3)) + k * loss \
            + (1 / 2) * torch.sum(p * p, dim=(1, 2, 3)) * m ** (-1)         # :697

        xp = x.detach().clone().requires_grad_(True)
        p = p - (epsilon / 2) * (xp.detach() + k * g)           # :702
        for _ in range(L):                                      # :704-713
            xp = xp + epsilon * m ** (-1) * p
            xp = xp.detach().requires_grad_(True)
            xt, loss, _, g = _data_loss_and_grad(xp, b, seq, seq_next, model, Hop, y_0)
            p = p - epsilon * (xp.detach() + k * g)
        p = p + (epsilon / 2) * (xp.detach() + k * g)           # :715

        H1 = (1 / 2) * torch.sum(xp.detach() ** 2, dim=(1, 2, 3)) + k * loss \
            + (1 / 2) * torch.sum(p * p, dim=(1, 2, 3)) * m ** (-1)         # :717
        dH = H1 - H0
        ratio = min(torch.tensor([1.0]), torch.exp(-dH))        # :719 (batch 1 only)
        u = torch.rand(1).item()                                # :720, CPU generator
        accept = u < ratio.item()
        if trace is not None:
            trace.setdefault('dH', []).append(float(dH.item()))
            trace.setdefault('accept', []).append(bool(accept))
            trace.setdefault('sigma_y', []).append(float(sigma_y))
            trace.setdefault('eps', []).append(float(epsilon))
        if accept:
            rejected = 0
            if epoch >= epochs + sampling:                      # :724-726, pre-increment epoch
                finals.append(xt.clone()[0])
            epoch += 1
            x = xp.detach().clone().requires_grad_(True)
            if trace is not None:
                trace.setdefault('psnr', []).append(
                    float(psnr_unit(to_unit_range(xt[0]), orig01[0]).item()))
        else:
            rejected += 1
            if rejected >= 2:                                   # :743-749
                tau = tau * 0.95
                epsilon = epsilon * 0.95
    return torch.stack(finals)


# --------------------------------------------------------------------------- #
# per-chain forms
# --------------------------------------------------------------------------- #

def _col(v, B):
    """fp64 per-chain scalars -> fp32 [B,1,1,1] (the rounding `scalar * tensor` applies)."""
    a = np.broadcast_to(np.asarray(v, dtype=np.float64), (B,))
    return torch.from_numpy(a.astype(np.float32)).view(B, 1, 1, 1)


def hamiltonian(Sx, loss, Sp, k, m):
    """main_sampling.py:697 / :717 per chain, in the reference's fp32 op order:
    (0.5*Sx + k*loss) + (0.5*Sp) * m^-1."""
    kf = torch.from_numpy(np.asarray(k, dtype=np.float64).astype(np.float32)).reshape(-1)
    minv = np.float32(m ** (-1))
    return (0.5 * Sx + kf * loss) + (0.5 * Sp) * minv


def trajectory(x, p, b, seq, seq_next, model, Hop, y_0, *, sigma_y, eps, m, L):
    """One outer iteration for B independent chains (main_sampling.py:693-718).

    x, p : [B,C,H,W] fp32.  sigma_y, eps : float or length-B array (fp64).
    Returns dict(x, p, xt, loss, H0, H1) with loss/H per chain."""
    B = x.shape[0]
    sig = np.broadcast_to(np.asarray(sigma_y, dtype=np.float64), (B,))
    ep = np.broadcast_to(np.asarray(eps, dtype=np.float64), (B,))
    k64 = 1 / (2 * sig ** 2)
    kf, eh, ef, ex = _col(k64, B), _col(ep / 2, B), _col(ep, B), _col(ep * m ** (-1), B)

    x0 = x.detach().clone().requires_grad_(True)
    _, _, loss_b, g = _data_loss_and_grad(x0, b, seq, seq_next, model, Hop, y_0)
    Sx = torch.sum(x0.detach() ** 2, dim=(1, 2, 3))
    Sp = torch.sum(p * p, dim=(1, 2, 3))
    H0 = hamiltonian(Sx, loss_b, Sp, k64, m)

    xp = x0.detach().clone()
    p = p - eh * (xp + kf * g)
    xt = None
    for _ in range(L):
        xp = (xp + ex * p).detach().requires_grad_(True)
        xt, _, loss_b, g = _data_loss_and_grad(xp, b, seq, seq_next, model, Hop, y_0)
        xp = xp.detach()
        p = p - ef * (xp + kf * g)
    p = p + eh * (xp + kf * g)
    Sx = torch.sum(xp ** 2, dim=(1, 2, 3))
    Sp = torch.sum(p * p, dim=(1, 2, 3))
    H1 = hamiltonian(Sx, loss_b, Sp, k64, m)
    return dict(x=xp, p=p, xt=xt, loss=loss_b, H0=H0, H1=H1, grad=g)


def hmc_chains(x, b, seq, seq_next, model, Hop, y_0, x_orig, *, tau, epsilon, m, sigma_0,
               draw_p, draw_u, epochs=60, sampling=20, max_iters=100000, trace=None):
    """Per-chain generalisation of main_sampling.py:660-774.

    draw_p(it) -> [B,C,H,W] standard normals, draw_u(it) -> [B] uniforms in [0,1).
    Every chain has its own (epoch, sigma_y, eps, tau, rejected); a chain whose
    epoch reached epochs + 2*sampling is frozen.  Returns [B, sampling, C, H, W]."""
    B = x.shape[0]
    L = max(1, math.floor(tau / epsilon))
    total = epochs + 2 * sampling
    epoch = np.zeros(B, dtype=np.int64)
    rejected = np.zeros(B, dtype=np.int64)
    tau_b = np.full(B, tau, dtype=np.float64)
    eps_b = np.full(B, epsilon, dtype=np.float64)
    sig_b = np.zeros(B, dtype=np.float64)
    x = x.detach().clone()
    finals = [[] for _ in range(B)]
    it = 0
    while (epoch < total).any() and it < max_iters:
        active = epoch < total
        for c in range(B):
            if not active[c]:
                continue
            if epoch[c] < epochs:
                sig_b[c] = sigma_0 + 1.6 * (1 - epoch[c] / epochs) ** 2
            elif epoch[c] == epochs:
                sig_b[c] = sigma_0
                if tau_b[c] > 0.1:
                    tau_b[c] = 0.1
                    eps_b[c] = 0.01
        p = draw_p(it) * math.sqrt(m)
        out = trajectory(x, p, b, seq, seq_next, model, Hop, y_0,
                         sigma_y=np.where(active, sig_b, 1.0), eps=np.where(active, eps_b, 0.0),
                         m=m, L=L)
        dH = out['H1'] - out['H0']
        ratio = torch.clamp(torch.exp(-dH), max=1.0)
        u = draw_u(it)
        acc = (u < ratio).numpy() & active
        if trace is not None:
            trace.setdefault('dH', []).append(dH.numpy().copy())
            trace.setdefault('accept', []).append(acc.copy())
            trace.setdefault('epoch', []).append(epoch.copy())
        for c in range(B):
            if not active[c]:
                continue
            if acc[c]:
                rejected[c] = 0
                if epoch[c] >= epochs + sampling:
                    finals[c].append(out['xt'][c].clone())
                epoch[c] += 1
                x[c] = out['x'][c]
            else:
                rejected[c] += 1
                if rejected[c] >= 2:
                    tau_b[c] *= 0.95
                    eps_b[c] *= 0.95
        it += 1
    return torch.stack([torch.stack(f) for f in finals])


# --------------------------------------------------------------------------- #
# single-step forms (what one fused kernel launch covers)
# --------------------------------------------------------------------------- #

def leapfrog_update(mode, x, p, g, *, eps, sigma_y, m):
    """The three fused-update variants, in the reference's op order.

    'first': Sx,Sp of the inputs (:697); p -= (eps/2)(x + k g) (:702); x += (eps/m) p (:706)
    'mid'  : p -= eps (x + k g) (:713);  x += (eps/m) p (:706, next iteration)
    'last' : p -= eps (x + k g) (:713);  p += (eps/2)(x + k g) (:715);  Sx,Sp of the outputs (:717)
    Returns (x, p, Sx, Sp) with Sx/Sp fp64 per-chain sums of squares (None for 'mid')."""
    B = x.shape[0]
    sig = np.broadcast_to(np.asarray(sigma_y, dtype=np.float64), (B,))
    ep = np.broadcast_to(np.asarray(eps, dtype=np.float64), (B,))
    kf, eh, ef, ex = _col(1 / (2 * sig ** 2), B), _col(ep / 2, B), _col(ep, B), _col(ep * m ** (-1), B)
    sq = lambda t: (t.double() ** 2).reshape(B, -1).sum(1)
    G = x + kf * g
    if mode == 'first':
        Sx, Sp = sq(x), sq(p)
        p = p - eh * G
        return x + ex * p, p, Sx, Sp
    if mode == 'mid':
        p = p - ef * G
        return x + ex * p, p, None, None
    p = p - ef * G
    p = p + eh * G
    return x, p, sq(x), sq(p)


def data_term(xt, Hop, y_0, apply_clip=True):
    """loss per chain and d(sum loss)/d xt through the final clip (:693-695), via autograd."""
    leaf = xt.detach().clone().requires_grad_(True)
    z = leaf.clip(-1, 1) if apply_clip else leaf
    resid = y_0 - Hop.H(z)
    per_chain = (resid ** 2).reshape(resid.shape[0], -1).sum(1)
    grad = torch.autograd.grad(per_chain.sum(), leaf)[0]
    return per_chain.detach(), grad
