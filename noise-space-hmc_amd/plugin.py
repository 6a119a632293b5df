"""The reference's algo-plugin surface, with the DDIM step on the HIP kernels.

`Base_Algo` keeps the constructor and the two hooks of algos/base_algo.py:3-16.  `HMC` is the
plugin `--algo hmc` uses: the reference dispatches that name to `Unconditional`
(main_sampling.py:254-255, algos/unconditional.py:4-28), whose `cal_x0` / `map_back` it re-implements
as differentiable ops backed by `nhmc_ddim_mix_fwd` / `nhmc_ddim_mix_bwd` / `nhmc_ddim_map_back`.

Two ways in:
  * the plugin surface  cal_x0(xt, t, at, at_next, y_0, noise) -> (x0_t, add_up), then
    map_back(x0_t, y_0, add_up, at_next, at) -> xt_next : same signatures and autograd behaviour as
    the reference, usable by its `iterative_sampling` loop unchanged;
  * `fused_step` / `fused_step_vjp`: the whole DDIM step in one kernel each way (3T / 5T of HBM
    traffic instead of the surface's 4T + 3T), which the sampler's own decode uses.
"""
from abc import ABC, abstractmethod

import torch

from . import kernels as K


class Base_Algo(ABC):
    def __init__(self, model, H_funcs, sigma_0, cls_fn=None):
        self.model = model
        self.H_funcs = H_funcs
        self.sigma_0 = sigma_0
        self.cls_fn = cls_fn

    @abstractmethod
    def cal_x0(self, xt, t, at, at_next, classes):
        ...

    @abstractmethod
    def map_back(self, x0_t, y_0, add_up, at_next, at):
        ...


class _PredictX0(torch.autograd.Function):
    """(xt, e) -> (x0_t, add_up) = (clip((xt - e sqrt(1-at))/sqrt(at)), sqrt(1-at_next) e[:, :C])."""

    @staticmethod
    def forward(ctx, xt, e, at, at_next):
        xt_c, e_c = xt.contiguous(), e.contiguous()
        out = K.ddim_mix_fwd(xt_c, e_c, at, at_next, want=('x0_t', 'add_up'))
        ctx.save_for_backward(xt_c, e_c, at, at_next)
        return out['x0_t'], out['add_up']

    @staticmethod
    def backward(ctx, g_x0, g_add):
        xt, e, at, at_next = ctx.saved_tensors
        if g_x0 is None:
            g_x0 = torch.zeros_like(xt)
        if g_add is None:
            g_add = torch.zeros_like(xt)
        g_xt, g_e = K.ddim_mix_bwd(g_add.contiguous(), xt, e, at, at_next, g_x0=g_x0.contiguous())
        return g_xt, g_e, None, None


class _MapBack(torch.autograd.Function):
    """(x0_t, add_up) -> sqrt(at_next) x0_t + add_up."""

    @staticmethod
    def forward(ctx, x0_t, add_up, at_next):
        ctx.save_for_backward(at_next)
        return K.ddim_map_back(x0_t.contiguous(), add_up.contiguous(), at_next)

    @staticmethod
    def backward(ctx, gout):
        (at_next,) = ctx.saved_tensors
        gout = gout.contiguous()
        return K.ddim_map_back(gout, torch.zeros_like(gout), at_next), gout, None


class HMC(Base_Algo):
    """Drop-in for `Unconditional` (algos/unconditional.py) under `--algo hmc`."""

    def __init__(self, model, H_funcs, sigma_0, cls_fn=None):
        super().__init__(model, H_funcs, sigma_0, cls_fn)
        if cls_fn is not None:
            raise NotImplementedError('classifier guidance (cls_fn) is outside the HMC hot path')

    def score(self, xt, t):
        """algos/unconditional.py:12 -- the PyTorch-ROCm score network; [B, C or 2C, H, W]."""
        return self.model(xt, t)

    @torch.enable_grad()
    def cal_x0(self, xt, t, at, at_next, y_0=None, noise='ddpm', classes=None):
        et = self.score(xt, t)
        self.et = et[:, :xt.shape[1]] if et.size(1) == 2 * xt.shape[1] else et     # reference stores the sliced score
        return _PredictX0.apply(xt, et, at, at_next)

    def map_back(self, x0_t, y_0, add_up, at_next, at):
        return _MapBack.apply(x0_t, add_up, at_next)

    # ---- fused forms used by nhmc.sampler -----------------------------------------------------
    @staticmethod
    def fused_step(xt, e, at, at_next, final_clip=False):
        return K.ddim_mix_fwd(xt, e, at, at_next, final_clip=final_clip)['xt_next']

    @staticmethod
    def fused_step_vjp(gout, xt, e, at, at_next, final_clip=False, gout2=None):
        return K.ddim_mix_bwd(gout, xt, e, at, at_next, final_clip=final_clip, gout2=gout2)


class HMCLatent(HMC):
    """Drop-in for `Unconditional_Latent` (algos/unconditional_latent.py) under `--algo hmc_latent`: the score is
    the latent-diffusion model's `apply_model(xt, t, None)` (:12); the DDIM arithmetic is the same kernels."""

    def score(self, xt, t):
        return self.model.apply_model(xt, t, None)
