"""Oracle: the latent-diffusion model object as the reference's `hmc_latent` sees it (test infrastructure, see
oracle/__init__.py; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

The reference builds `ldm.models.diffusion.ddpm.LatentDiffusion` (needs pytorch_lightning + taming, absent offline) and
touches four members of it (main_sampling_latent.py:651,670,771-772; algos/unconditional_latent.py:12).  This file
restates those four on the CPU around ANY pair of torch networks:

  * `vq_straight_through`   taming-transformers==0.0.1 (environment.yml:24, not vendored) `VectorQuantizer2.forward`,
                            restated from its published source -- "parity unpinned" for the quantiser itself (no
                            reference file or fixture holds its outputs); anchored on the call site
                            ldm/models/autoencoder.py:274-279 (`quant, emb_loss, info = self.quantize(h)`).
  * `alphas_cumprod_ldm`    ldm/models/diffusion/ddpm.py:117-138 (`register_schedule`) over `make_beta_schedule`
                            (ldm/modules/diffusionmodules/util.py:21-25), pinned by G12 against that function.
  * `OracleLatent`          apply_model (ddpm.py:892-893: `@torch.no_grad()` in this repository, so the score carries
                            no gradient), differentiable_decode_first_stage (ddpm.py:766-820 ->
                            autoencoder.py:274-283: quantise, post_quant_conv, decoder), alphas_cumprod(_prev).

In gen_golden_ldm.py the two networks are the REFERENCE's own classes (openaimodel.UNetModel, model.Decoder); in the
GPU parity tests they are nhmc.ldm's modules on the CPU, which G12 pins to those classes.
"""
import hashlib
import math

import numpy as np
import torch


def vq_straight_through(z, codebook):
    """VectorQuantizer2.forward (legacy=True, remap=None), value and gradient: z [B,D,h,w], codebook [n,D]."""
    zl = z.permute(0, 2, 3, 1).contiguous()                                   # b c h w -> b h w c
    flat = zl.view(-1, codebook.shape[1])
    d = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) \
        - 2 * torch.einsum('bd,dn->bn', flat, codebook.t())
    idx = torch.argmin(d, dim=1)
    zq = codebook[idx].view(zl.shape)
    zq = zl + (zq - zl).detach()                                              # straight-through
    return zq.permute(0, 3, 1, 2).contiguous(), idx.view(zl.shape[:3])


def alphas_cumprod_ldm(timesteps=1000, linear_start=1e-4, linear_end=2e-2):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1. - betas, axis=0)
    prev = np.append(1., ac[:-1])
    return torch.tensor(ac, dtype=torch.float32), torch.tensor(prev, dtype=torch.float32)


class F64Net:
    """One of nhmc's torch networks evaluated in float64 and rounded to float32 (inputs and outputs stay fp32): removes
    the CPU-vs-GPU convolution rounding noise that a 70-trajectory comparison through Metropolis decisions would
    otherwise amplify.  The sinusoidal timestep embedding is evaluated in float64 too (a forward pre-hook on
    `time_embed` substitutes it), because fp32 sin/cos differ by an ulp between hosts and devices."""

    def __init__(self, net):
        import copy
        self.net = copy.deepcopy(net).double()
        te = getattr(self.net, 'time_embed', None)
        self.emb_dim = te[0].in_features if te is not None else None
        if te is not None:
            te.register_forward_pre_hook(lambda mod, args: (self._emb,))

    def to(self, device):
        self.net = self.net.to(device)
        return self

    def __call__(self, *args):
        if self.emb_dim is not None:
            t = args[1].double()
            half = self.emb_dim // 2
            freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float64, device=t.device) / half)
            ang = t[:, None] * freqs[None]
            self._emb = torch.cat([ang.cos(), ang.sin()], dim=-1)
        return self.net(*[a.double() for a in args]).float()


class OracleLatent:
    def __init__(self, unet, decoder, post_quant_conv, codebook, *, linear_start, linear_end, timesteps=1000,
                 scale_factor=1.0, score_no_grad=True):
        self.unet, self.decoder, self.post_quant_conv = unet, decoder, post_quant_conv
        self.codebook = codebook.detach()
        self.alphas_cumprod, self.alphas_cumprod_prev = alphas_cumprod_ldm(timesteps, linear_start, linear_end)
        self.scale_factor, self.score_no_grad = scale_factor, score_no_grad

    def apply_model(self, x_noisy, t, cond=None):
        if self.score_no_grad:
            with torch.no_grad():
                return self.unet(x_noisy, t)
        return self.unet(x_noisy, t)

    def differentiable_decode_first_stage(self, z):
        z = 1. / self.scale_factor * z
        quant, _ = vq_straight_through(z, self.codebook)
        return self.decoder(self.post_quant_conv(quant))

    @classmethod
    def from_product(cls, model, f64=False):
        """CPU twin of an nhmc.ldm.LatentDiffusion (same torch networks, quantiser restated above)."""
        import copy
        m = copy.deepcopy(model).cpu()
        fs = m.first_stage_model
        wrap = F64Net if f64 else (lambda net: net)
        return cls(wrap(m.model.diffusion_model), wrap(torch.nn.Sequential(fs.post_quant_conv, fs.decoder)), (lambda q: q),
                   fs.quantize.embedding.weight,
                   linear_start=0.0, linear_end=0.0, scale_factor=m.scale_factor,
                   score_no_grad=not m.differentiable_score)._with_alphas(m.alphas_cumprod, m.alphas_cumprod_prev)

    def _with_alphas(self, ac, prev):
        self.alphas_cumprod, self.alphas_cumprod_prev = ac.clone(), prev.clone()
        return self


def seeded_state(sd, seed):
    """Deterministic non-degenerate weights from the key order and shapes alone (tests rebuild the same tensors):
    matrices / filters ~ N(0, 1/fan_in), norm scales 1 + 0.1 N, everything else 0.05 N."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        r = torch.randn(v.shape, generator=g)
        if v.dim() >= 2:
            out[k] = r / math.sqrt(v[0].numel())
        elif 'norm' in k and k.endswith('weight') or k.endswith('.0.weight') and v.dim() == 1:
            out[k] = 1 + 0.1 * r
        else:
            out[k] = 0.05 * r
    return out


def keys_hash(sd, prefix=''):
    text = '\n'.join(f'{prefix}{k} {tuple(v.shape)}' for k, v in sd.items())
    return hashlib.sha256(text.encode()).hexdigest()
