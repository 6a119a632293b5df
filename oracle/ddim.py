"""Oracle: deterministic DDIM decode (test infrastructure, see oracle/__init__.py).

Restates
  * `Unconditional.cal_x0`   algos/unconditional.py:9-24
  * `Unconditional.map_back` algos/unconditional.py:26-28
  * `iterative_sampling`     main_sampling.py:898-915
and, for the kernels' backward, the closed-form VJP autograd derives from them
(SURVEY.md section 8 row a11).
"""
import numpy as np
import torch

from .schedule import alpha_bar


def sqrt_rn(t):
    """IEEE correctly rounded fp32 square root of a (grad-free) tensor.

    torch's CPU `sqrt` is NOT correctly rounded and depends on the host CPU: against the exact
    root it is 1 ulp off for 6 of the 1001 alpha-bar table entries in the build container and for
    183 of them on the GPU box's host (both "AVX512" builds; measured), while numpy's fp32 sqrt,
    torch's GPU sqrt and the HIP kernels' sqrtf are exact everywhere.  The oracle therefore takes
    the root with numpy so that it means the same thing on every machine; for the alpha-bars the
    reference path uses (t = 750/500/250/-1) the container's torch sqrt is exact too, so the
    fixtures captured from the reference are unaffected (tests/test_oracle_golden.py)."""
    return torch.from_numpy(np.sqrt(t.detach().numpy()))


def predict_x0(xt, et, at, at_next):
    """algos/unconditional.py:17-24 after the score call.  `et` may carry the
    learned-sigma channels (6); only the first 3 are used (:17-18).
    Division (not reciprocal multiply) and fp32 sqrt of [n,1,1,1] tensors."""
    if et.size(1) == 6:
        et = et[:, :3]
    x0_t = (xt - et * sqrt_rn(1 - at)) / sqrt_rn(at)
    x0_t = x0_t.clip(-1, 1)
    add_up = sqrt_rn(1 - at_next) * et
    return x0_t, add_up


def renoise(x0_t, add_up, at_next):
    """algos/unconditional.py:26-28."""
    return sqrt_rn(at_next) * x0_t + add_up


def ddim_step(xt, et, at, at_next):
    x0_t, add_up = predict_x0(xt, et, at, at_next)
    return renoise(x0_t, add_up, at_next)


def decode(x, b, seq, seq_next, model):
    """main_sampling.py:898-915 with the `Unconditional` plugin inlined.

    `model(xt, t)` gets float timesteps `ones(n) * i` (:903) and returns
    [n, 3 or 6, H, W]."""
    n = x.shape[0]
    xt = x
    for i, j in zip(reversed(seq), reversed(seq_next)):
        t = torch.ones(n) * i
        nt = torch.ones(n) * j
        at = alpha_bar(b, t.long())
        at_next = alpha_bar(b, nt.long())
        et = model(xt, t)
        xt = ddim_step(xt, et, at, at_next)
    return xt


def ddim_step_vjp(gout, xt, et, at, at_next):
    """Hand-derived VJP of `ddim_step` w.r.t. (xt, et[:, :3]) in autograd's op order.

    With c1 = sqrt(1-at), c2 = sqrt(at), c3 = sqrt(at_next), c4 = sqrt(1-at_next),
    u = (xt - et c1)/c2, mask = 1[-1 <= u <= 1]:
        d/dxt = ((gout c3) mask) / c2
        d/det = gout c4 - (d/dxt) c1            (zero for channels 3..5)
    """
    e3 = et[:, :3]
    c1, c2 = sqrt_rn(1 - at), sqrt_rn(at)
    c3, c4 = sqrt_rn(at_next), sqrt_rn(1 - at_next)
    u = (xt - e3 * c1) / c2
    mask = ((u >= -1) & (u <= 1)).to(gout.dtype)
    g_u = (gout * c3) * mask / c2
    g_e3 = c4 * gout + (-g_u) * c1
    g_e = torch.zeros_like(et)
    g_e[:, :3] = g_e3
    return g_u, g_e
