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


def round_bits(t, bits):
    """float64 tensor -> the nearest value with `bits` significant bits (ties to even).  frexp / round / ldexp are exact
    operations, so the result is the same on any IEEE device whenever the inputs agree to better than the spacing."""
    m, e = torch.frexp(t)                                      # t = m 2^e, 0.5 <= |m| < 1
    return torch.ldexp(torch.round(torch.ldexp(m, torch.full_like(e, bits))), e - bits)


def round_grid(t, bits):
    """float64 tensor -> the nearest multiple of q = 2^(e - bits), 2^e the power of two just above max |t| (ties to even).
    An ABSOLUTE grid per tensor: an entry that is small because its terms cancel (where the CPU's and the GPU's float64
    results differ by far more than 1e-16 of the entry) is rounded as coarsely as the large ones, which `round_bits`
    (a relative grid per entry) does not do.  Exact operations; values stay exact in fp32 for bits <= 24."""
    _, e = torch.frexp(t.abs().max())
    q = torch.ldexp(torch.ones((), dtype=t.dtype, device=t.device), e - bits)
    return torch.round(t / q) * q


class _GridNet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, net, bits, absolute):
        with torch.enable_grad():
            xd = x.detach().double().requires_grad_(True)
            out = net(xd, t.double())
        ctx.xd, ctx.out, ctx.bits, ctx.absolute = xd, out, bits, absolute
        return (round_grid if absolute else round_bits)(out.detach(), bits).float()

    @staticmethod
    def backward(ctx, g):
        (gin,) = torch.autograd.grad(ctx.out, ctx.xd, g.double())
        return (round_grid if ctx.absolute else round_bits)(gin, ctx.bits).float(), None, None, None, None


class GridF64Score(nn.Module):
    """F64Score made REPRODUCIBLE ACROSS DEVICES.  A float64 network still differs between the CPU and the GPU in its last
    bits (libm tanh, convolution summation order), and when such a value sits within ~1e-16 of an fp32 rounding boundary
    the fp32 result flips: measured on the 256 x 256 runs, about 3 of every 1e9 score outputs (tools/trace_replay.py;
    a flip under a local operator is mostly absorbed by a clip or a rounding, under a global transform like the
    Walsh-Hadamard operator it is spread over the whole gradient and the replay leaves the reference's run for good).
    Here the output -- and, through a custom autograd function, the input gradient -- is rounded in float64 to `bits`
    significant bits before the conversion: every such value is exact in fp32, so the conversion no longer rounds, and a
    flip needs the float64 value within ~1e-16 of a midpoint of a 2^-bits grid, 2^(24 - bits) times rarer (bits = 10:
    once in ~1e12 values, ~0.03 expected per whole run).  Still an ARGUMENT of the reference's `hmc()`."""

    def __init__(self, net, bits=10, absolute=False):
        """absolute: round on one absolute grid per tensor (`round_grid`, bits below the tensor's largest magnitude)
        instead of per-entry significant bits -- for problems whose input gradient has entries that are small by
        cancellation (the box mask: the gradient vanishes inside the box)."""
        super().__init__()
        import copy
        self.net, self.bits, self.absolute = copy.deepcopy(net).double(), bits, bool(absolute)

    def forward(self, x, t):
        return _GridNet.apply(x, t, self.net, self.bits, self.absolute)
