"""Latent-diffusion model object for `--algo hmc_latent` (BASELINE configs[4]): LDM U-Net + VQ-f4 first stage.

The reference instantiates `ldm.models.diffusion.ddpm.LatentDiffusion` from configs/config_ffhq_latent.yml:33-80
(main_sampling_latent.py:124-127, ldm_loader.py:11-25); that class needs pytorch_lightning + taming, and the latent
sampler touches four things of it only (main_sampling_latent.py:651,670,771-772; algos/unconditional_latent.py:12):

    apply_model(x, t, cond)                  -> ldm/models/diffusion/ddpm.py:892-893,... -> DiffusionWrapper ->
                                                openaimodel.UNetModel.forward (openaimodel.py:710-741)
    differentiable_decode_first_stage(z)     -> ddpm.py:766-820 -> VQModelInterface.decode (autoencoder.py:274-283):
                                                quantize -> post_quant_conv -> model.Decoder (model.py:462-565)
    alphas_cumprod, alphas_cumprod_prev      -> register_schedule (ddpm.py:117-138)

This file is an own PyTorch-ROCm statement of those four, with the module tree and parameter names of the
checkpoints the reference loads (`models/ldm/model.ckpt`: `model.diffusion_model.*`, `first_stage_model.decoder.*`,
`first_stage_model.post_quant_conv.*`, `first_stage_model.quantize.embedding.weight`), so `load_checkpoint` takes
them unchanged.  Like nhmc.unet it is plumbing for the path (north_star keeps the networks on PyTorch-ROCm); the one
piece with a HIP kernel is the codebook lookup (`nhmc_vq_nearest`): the torch form materialises an
[n_pixels, n_embed] distance matrix (2.1 GB at 16 chains) where the kernel keeps the codebook in LDS.

Two behaviours of the reference that are easy to miss and are kept:
  * `LatentDiffusion.apply_model` is decorated `@torch.no_grad()` in this repository (ddpm.py:892, "only for inverse
    problem solving"): the score is a constant of the decode, the data-term gradient reaches the latent through the
    DDIM mix only.  `differentiable_score=True` lifts that (an extension, off by default).
  * the first stage quantises on decode (`force_not_quantize=False`, ddpm.py:817): nearest codebook entry forward,
    straight-through gradient (taming-transformers 0.0.1, `VectorQuantizer2.forward`: z_q = z + (z_q - z).detach()).
"""

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from .unet import AttentionBlock, Stage, _BiasAdd2, conv_nobias, fused_glue, group_norm_act, group_norm_act_fork, sinusoid, sinusoid_freqs

# configs/config_ffhq_latent.yml:45-80
FFHQ_LDM_UNET = dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, attention_resolutions=(8, 4, 2),
                     num_res_blocks=2, channel_mult=(1, 2, 3, 4), num_head_channels=32)
FFHQ_VQ_F4 = dict(embed_dim=3, n_embed=8192,
                  ddconfig=dict(double_z=False, z_channels=3, resolution=256, in_channels=3, out_ch=3, ch=128,
                                ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(), dropout=0.0))
FFHQ_LDM = dict(linear_start=0.0015, linear_end=0.0195, timesteps=1000, unet_config=FFHQ_LDM_UNET,
                first_stage_config=FFHQ_VQ_F4)


# --------------------------------------------------------------------------------------------- #
# U-Net (openaimodel.UNetModel with the options the LDM configs use: additive time embedding, learned resampling)
# --------------------------------------------------------------------------------------------- #
class ConvDown(nn.Module):
    """openaimodel.py:126-154 with use_conv: stride-2 3x3 convolution, parameter name `op`."""

    def __init__(self, ch):
        super().__init__()
        self.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.op(x)


class ConvUp(nn.Module):
    """openaimodel.py:88-114 with use_conv: nearest x2 then 3x3 convolution, parameter name `conv`."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='nearest'))


class AddEmbResBlock(nn.Module):
    """openaimodel.py:157-267 without scale-shift norm: h = conv(silu(gn(x))) + W emb; out = skip(x) + conv(silu(gn(h)))."""
    takes_emb = True

    def __init__(self, ch, emb_ch, out_ch):
        super().__init__()
        self.in_layers = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(ch, out_ch, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, out_ch))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_ch), nn.SiLU(), nn.Identity(), nn.Conv2d(out_ch, out_ch, 3, padding=1))
        self.skip_connection = nn.Identity() if out_ch == ch else nn.Conv2d(ch, out_ch, 1)

    def forward(self, x, emb):
        conv1, conv2, e = self.in_layers[2], self.out_layers[3], self.emb_layers(emb)
        h = group_norm_act(self.in_layers[0], x)
        if fused_glue(h, conv1.bias, conv2.bias, e):
            # conv1's bias and the embedding term (h = conv(..) + emb_out, openaimodel.py:257) enter the next GroupNorm's
            # load as one [B, C] term; conv2's bias enters the residual add
            h = group_norm_act(self.out_layers[0], conv_nobias(conv1, h), pre=(e + conv1.bias).contiguous())
            return _BiasAdd2.apply(conv_nobias(conv2, h), conv2.bias, self.skip_connection(x))
        h = conv1(h) + e[:, :, None, None]
        return self.skip_connection(x) + conv2(group_norm_act(self.out_layers[0], h))


class LDMUNet(nn.Module):
    """Module tree of ldm/modules/diffusionmodules/openaimodel.py:413-741 for conv_resample, legacy attention order,
    no class / context conditioning (the FFHQ latent config).  `attention_resolutions` are downsampling factors."""

    def __init__(self, image_size=64, in_channels=3, model_channels=224, out_channels=3, num_res_blocks=2,
                 attention_resolutions=(8, 4, 2), channel_mult=(1, 2, 3, 4), num_head_channels=32, **other):
        super().__init__()
        # every other constructor option of the reference class must sit at the value the FFHQ latent config leaves it
        neutral = dict(dropout=0, conv_resample=True, dims=2, num_classes=None, use_fp16=False, num_heads=-1,
                       num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                       use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                       context_dim=None, n_embed=None, legacy=True)
        other.pop('use_checkpoint', None)                                   # activation checkpointing: no arithmetic effect
        bad = sorted(k for k, v in other.items() if k not in neutral or neutral[k] != v)
        if bad or num_head_channels <= 0:
            raise NotImplementedError(f'LDMUNet: options outside the FFHQ latent config: {bad or "num_head_channels"}')
        self.model_channels = mc = model_channels
        self.in_channels, self.out_channels, self.image_size = in_channels, out_channels, image_size
        emb = 4 * mc
        self.time_embed = nn.Sequential(nn.Linear(mc, emb), nn.SiLU(), nn.Linear(emb, emb))
        self.register_buffer('_freqs', sinusoid_freqs(mc), persistent=False)
        att = tuple(int(a) for a in attention_resolutions)
        ch, ds, skips = mc, 1, [mc]
        self.input_blocks = nn.ModuleList([Stage(nn.Conv2d(in_channels, mc, 3, padding=1))])
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [AddEmbResBlock(ch, emb, mult * mc)]
                ch = mult * mc
                if ds in att:
                    layers.append(AttentionBlock(ch, num_head_channels))
                self.input_blocks.append(Stage(*layers))
                skips.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(Stage(ConvDown(ch)))
                skips.append(ch)
                ds *= 2
        self.middle_block = Stage(AddEmbResBlock(ch, emb, ch), AttentionBlock(ch, num_head_channels), AddEmbResBlock(ch, emb, ch))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [AddEmbResBlock(ch + skips.pop(), emb, mc * mult)]
                ch = mc * mult
                if ds in att:
                    layers.append(AttentionBlock(ch, num_head_channels))
                if level and i == num_res_blocks:
                    layers.append(ConvUp(ch))
                    ds //= 2
                self.output_blocks.append(Stage(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(mc, out_channels, 3, padding=1))

    def forward(self, x, timesteps=None, context=None, y=None):
        emb = self.time_embed(sinusoid(timesteps, self.model_channels, self._freqs))
        hs, h = [], x
        for blk in self.input_blocks:
            h = blk(h, emb)
            hs.append(h)
        h = self.middle_block(h, emb)
        for blk in self.output_blocks:
            h = blk(torch.cat([h, hs.pop()], dim=1), emb)
        return self.out[2](group_norm_act(self.out[0], h))


# --------------------------------------------------------------------------------------------- #
# VQ-f4 first stage, decode side (ldm/modules/diffusionmodules/model.py)
# --------------------------------------------------------------------------------------------- #
def _gn(ch):
    return nn.GroupNorm(32, ch, eps=1e-6, affine=True)                     # model.py:38-39


def _swish(x):
    return x * torch.sigmoid(x)                                            # model.py:33-35


class PlainResBlock(nn.Module):
    """model.py:82-141 with temb_channels = 0, dropout 0, 1x1 shortcut."""

    def __init__(self, cin, cout):
        super().__init__()
        self.norm1, self.conv1 = _gn(cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = _gn(cout), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h, x = group_norm_act_fork(self.norm1, x, act_fn=_swish)         # Normalize + x sigmoid(x); x goes on to the shortcut
        sc = self.nin_shortcut(x) if hasattr(self, 'nin_shortcut') else x
        if fused_glue(h, self.conv1.bias, self.conv2.bias):
            h = group_norm_act(self.norm2, conv_nobias(self.conv1, h), act_fn=_swish, pre=self.conv1.bias)
            return _BiasAdd2.apply(conv_nobias(self.conv2, h), self.conv2.bias, sc)
        h = self.conv2(group_norm_act(self.norm2, self.conv1(h), act_fn=_swish))
        return sc + h


class SpatialSelfAttention(nn.Module):
    """model.py:150-203: single-head attention over the h*w positions with 1x1 conv projections."""

    def __init__(self, ch):
        super().__init__()
        self.norm = _gn(ch)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(ch, ch, 1) for _ in range(4))

    def forward(self, x):
        h = group_norm_act(self.norm, x, act=False)
        b, c, hh, ww = x.shape
        q, k, v = (f(h).reshape(b, c, hh * ww) for f in (self.q, self.k, self.v))
        w = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** (-0.5)), dim=2)
        out = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
        return x + self.proj_out(out)


class _Level(nn.Module):
    pass


class UpConv(nn.Module):
    """model.py:42-58."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class VQDecoder(nn.Module):
    """model.py:462-565 (`Decoder`): conv_in -> mid (res, attn, res) -> per level (num_res_blocks+1 res blocks,
    upsample) from the coarsest level up -> norm, swish, conv_out."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels=None, resolution, z_channels, double_z=False, **ignored):
        super().__init__()
        if dropout or not resamp_with_conv:
            raise NotImplementedError('VQDecoder: dropout / conv-less resampling are outside the vq-f4 config')
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        res = resolution // 2 ** (self.num_resolutions - 1)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = PlainResBlock(block_in, block_in)
        self.mid.attn_1 = SpatialSelfAttention(block_in)
        self.mid.block_2 = PlainResBlock(block_in, block_in)
        levels = []
        for i_level in reversed(range(self.num_resolutions)):
            lvl = _Level()
            lvl.block, lvl.attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                lvl.block.append(PlainResBlock(block_in, block_out))
                block_in = block_out
                if res in attn_resolutions:
                    lvl.attn.append(SpatialSelfAttention(block_in))
            if i_level != 0:
                lvl.upsample = UpConv(block_in)
                res *= 2
            levels.insert(0, lvl)
        self.up = nn.ModuleList(levels)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                h = lvl.block[i_block](h)
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block](h)
            if i_level != 0:
                h = lvl.upsample(h)
        return self.conv_out(group_norm_act(self.norm_out, h, act_fn=_swish))


class _StraightThroughVQ(torch.autograd.Function):
    """Forward: z + (e[argmin_k |z - e_k|^2] - z) per latent pixel (HIP kernel); backward: identity."""

    @staticmethod
    def forward(ctx, z, codebook):
        zq, _idx = K.vq_nearest(z.contiguous(), codebook)
        return zq

    @staticmethod
    def backward(ctx, g):
        return g, None


class Codebook(nn.Module):
    """taming `VectorQuantizer2` reduced to what decode uses: `embedding.weight` [n_embed, embed_dim]."""

    def __init__(self, n_embed, embed_dim):
        super().__init__()
        self.embedding = nn.Embedding(n_embed, embed_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_embed, 1.0 / n_embed)

    def forward(self, z):
        return _StraightThroughVQ.apply(z, self.embedding.weight.detach().contiguous())

    def indices(self, z):
        return K.vq_nearest(z.contiguous(), self.embedding.weight.detach().contiguous())[1]


class VQFirstStage(nn.Module):
    """Decode side of `VQModelInterface` (ldm/models/autoencoder.py:263-283)."""

    def __init__(self, embed_dim, n_embed, ddconfig, **ignored):
        super().__init__()
        self.embed_dim = embed_dim
        self.decoder = VQDecoder(**ddconfig)
        self.quantize = Codebook(n_embed, embed_dim)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig['z_channels'], 1)

    def decode(self, h, force_not_quantize=False):
        quant = h if force_not_quantize else self.quantize(h)
        return self.decoder(self.post_quant_conv(quant))


# --------------------------------------------------------------------------------------------- #
# the model object the latent sampler is handed
# --------------------------------------------------------------------------------------------- #
class _Wrapper(nn.Module):
    """`DiffusionWrapper` (ddpm.py): only there so the parameters are named `model.diffusion_model.*`."""

    def __init__(self, net):
        super().__init__()
        self.diffusion_model = net


def ldm_alphas_cumprod(timesteps=1000, linear_start=1e-4, linear_end=2e-2):
    """ddpm.py:117-138 with the "linear" schedule of ldm/modules/diffusionmodules/util.py:21-25: betas are the squares
    of a linspace of square roots, in float64; the cumulative product is numpy float64, stored as float32."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64, device='cpu') ** 2).numpy()
    ac = np.cumprod(1.0 - betas, axis=0)
    prev = np.append(1.0, ac[:-1])
    return torch.tensor(ac, dtype=torch.float32, device='cpu'), torch.tensor(prev, dtype=torch.float32, device='cpu')


class LatentDiffusion(nn.Module):
    def __init__(self, unet_config=None, first_stage_config=None, linear_start=1e-4, linear_end=2e-2, timesteps=1000,
                 scale_factor=1.0, differentiable_score=False, **ignored):
        super().__init__()
        self.model = _Wrapper(LDMUNet(**_params(unet_config or FFHQ_LDM_UNET)))
        self.first_stage_model = VQFirstStage(**_params(first_stage_config or FFHQ_VQ_F4))
        ac, prev = ldm_alphas_cumprod(timesteps, linear_start, linear_end)
        self.register_buffer('alphas_cumprod', ac)
        self.register_buffer('alphas_cumprod_prev', prev)
        self.scale_factor = float(scale_factor)
        self.differentiable_score = bool(differentiable_score)
        self.channels = self.model.diffusion_model.in_channels
        self.image_size = self.model.diffusion_model.image_size

    def apply_model(self, x_noisy, t, cond=None):
        """ddpm.py:892-893: evaluated under no_grad in the reference."""
        if cond is not None:
            raise NotImplementedError('conditioning is outside the unconditional FFHQ latent config')
        if self.differentiable_score:
            return self.model.diffusion_model(x_noisy, t)
        with torch.no_grad():
            return self.model.diffusion_model(x_noisy, t)

    def differentiable_decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:766-820 (no `split_input_params` in this config)."""
        if predict_cids:
            raise NotImplementedError('predict_cids is outside the HMC path')
        z = 1. / self.scale_factor * z
        return self.first_stage_model.decode(z, force_not_quantize=force_not_quantize)

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        return self.differentiable_decode_first_stage(z, predict_cids, force_not_quantize)

    def load_checkpoint(self, path):
        """`models/ldm/model.ckpt` (ldm_loader.py:11-16): a Lightning checkpoint {'state_dict': ...}.  Takes the keys
        this object owns; the rest (encoder, EMA copy, loss, training buffers) is reported, not loaded."""
        blob = torch.load(path, map_location='cpu', weights_only=True)
        sd = blob.get('state_dict', blob)
        mine = self.state_dict()
        take = {k: v for k, v in sd.items() if k in mine}
        missing = [k for k in mine if k not in take]
        if missing:
            raise KeyError(f'{path}: {len(missing)} keys of the latent model are missing, e.g. {missing[:3]}')
        self.load_state_dict(take)
        return sorted(set(sd) - set(take))


def _params(cfg):
    """Accepts the yaml form {'target': ..., 'params': {...}} or the bare parameter dict."""
    cfg = dict(cfg)
    return dict(cfg.get('params', cfg))


def create_latent_model(config=None, ckpt='models/ldm/model.ckpt', quiet=False, **overrides):
    """`load_model_from_config` (ldm_loader.py:11-25).  `config`: the `model.params` mapping of
    configs/config_ffhq_latent.yml (or None for the FFHQ defaults).  A missing checkpoint leaves the random
    initialisation in place with a notice, as nhmc.unet does."""
    import os
    prm = dict(_params(config or FFHQ_LDM))
    prm.update(overrides)
    fs = dict(_params(prm.get('first_stage_config') or FFHQ_VQ_F4))
    fs_ckpt = fs.pop('ckpt_path', None)
    prm['first_stage_config'] = fs
    model = LatentDiffusion(**prm)
    if ckpt and os.path.exists(ckpt):
        model.load_checkpoint(ckpt)
    elif not quiet:
        print(f'checkpoint {ckpt} not found / randomly initialised (first stage: {fs_ckpt})')
    return model.eval().requires_grad_(False)
