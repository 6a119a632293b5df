"""Oracle: a tiny seeded stand-in for the score network (test infrastructure).

The pretrained FFHQ checkpoint is not available offline, and the reference's own
fallback is random initialisation (guided_diffusion/unet_ffhq.py:87-90).  Golden
trajectories therefore use this small smooth network with the reference's call
convention `model(xt, t) -> [B, 6, H, W]` (algos/unconditional.py:12,17-18); its
weights are committed under tests/golden/tiny_score.pt so that the reference run
that produced a fixture and every later replay use identical parameters.
"""
import math

import torch
import torch.nn as nn


class TinyScore(nn.Module):
    def __init__(self, width=8, out_ch=6):
        super().__init__()
        self.inp = nn.Conv2d(3, width, 3, padding=1)
        self.temb = nn.Linear(4, width)
        self.mid = nn.Conv2d(width, width, 3, padding=1)
        self.out = nn.Conv2d(width, out_ch, 3, padding=1)

    def forward(self, x, t):
        ph = t.to(x.dtype)[:, None] * torch.tensor([1.0, 2.0, 3.0, 5.0], device=x.device, dtype=x.dtype) * (math.pi / 1000.0)
        emb = self.temb(torch.cat([ph.sin()[:, :2], ph.cos()[:, 2:]], dim=1))
        h = torch.tanh(self.inp(x) + emb[:, :, None, None])
        h = torch.tanh(self.mid(h)) + h
        return self.out(h)


def make_tiny_score(seed=1234):
    g = torch.Generator().manual_seed(seed)
    net = TinyScore()
    with torch.no_grad():
        for prm in net.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.35 / math.sqrt(max(1, prm[0].numel()))))
    return net.eval().requires_grad_(False)


class F64Score(nn.Module):
    """A score network evaluated in float64 and rounded to float32 (inputs / outputs stay fp32): CPU (oneDNN) and GPU
    convolutions then agree to ~1e-15 before the rounding, so a many-trajectory comparison is sensitive to the sampler's
    kernels and not to the amplification of conv-implementation noise.  The model is an ARGUMENT of the reference's
    `hmc()`, so wrapping it leaves the reference code untouched (fixtures G14)."""

    def __init__(self, net):
        super().__init__()
        import copy
        self.net = copy.deepcopy(net).double()

    def forward(self, x, t):
        return self.net(x.double(), t.double()).float()
