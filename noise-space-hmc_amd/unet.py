"""Score network for the FFHQ config: an ADM-style U-Net in plain PyTorch-ROCm.

Out of scope as a kernel target (north_star keeps the score evaluations on PyTorch-ROCm; SURVEY.md
section 8 row a16) but needed to run and measure the hot path end to end.  The module tree and
parameter names follow guided_diffusion's checkpoint layout (`time_embed.0`, `input_blocks.k.0.in_layers.0`,
`...emb_layers.1`, `...out_layers.3`, `...skip_connection`, `k.1.norm/qkv/proj_out`, `middle_block`,
`output_blocks`, `out.0/out.2`) so that `models/ffhq_10m.pt` loads with `load_state_dict` unchanged
(reference: guided_diffusion/unet_ffhq.py:25-91 `create_model`, :467-734 `UNetModel`; config
configs/config_ffhq.yml:17-35).  When the checkpoint is absent the weights stay randomly initialised,
as in the reference's own fallback (unet_ffhq.py:87-90).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K

FFHQ_CONFIG = dict(image_size=256, num_channels=128, num_res_blocks=1, channel_mult='', learn_sigma=True,
                   attention_resolutions='16', num_head_channels=64, use_scale_shift_norm=True,
                   resblock_updown=True)
_DEFAULT_MULT = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}


def sinusoid_freqs(dim, max_period=10000):
    """Computed on the CPU exactly as the reference does (guided_diffusion/nn.py:114-116)."""
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)


def sinusoid(t, dim, freqs=None):
    freqs = sinusoid_freqs(dim).to(t.device) if freqs is None else freqs
    ang = t[:, None].float() * freqs[None]
    emb = torch.cat([ang.cos(), ang.sin()], dim=-1)
    return F.pad(emb, (0, dim % 2))


class _GroupNormAct(torch.autograd.Function):
    """y = act(GroupNorm(x + pre) (1 + scale) + shift) on the fused HIP kernels (csrc/gn_act.hip): 2 reads + 1 write forward,
    4 reads + 1 write for the input gradient, nothing but x saved -- against 5R + 4W / 7R + 3W for the ATen op sequence
    the reference's GroupNorm32 + scale-shift + SiLU lowers to (unet_ffhq.py:310-321)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, film, pre, groups, eps, act):
        xc = x if x.is_contiguous() else x.contiguous()
        y, ws, splits = K.gn_act_fwd(xc, gamma, beta, groups, eps, act, film, pre)
        ctx.save_for_backward(xc, gamma, beta, ws)
        ctx.consts = (film, pre)                              # constants of the path (no gradient flows into them)
        ctx.meta = (groups, eps, act, splits)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, gamma, beta, ws = ctx.saved_tensors
        groups, eps, act, splits = ctx.meta
        film, pre = ctx.consts
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = K.gn_act_bwd(xc, dy, gamma, beta, groups, eps, act, film, ws, splits, pre)
        return dx, None, None, None, None, None, None, None


class _GroupNormActFork(torch.autograd.Function):
    """_GroupNormAct with x itself as a second output, for a block whose input feeds the GroupNorm AND the skip path
    (unet_ffhq.py:299-321: `h = self.in_layers(x)` ... `self.skip_connection(x) + h`).  The two gradients of x then meet
    inside the backward kernel (dx = fl(dx_groupnorm) + dx_skip, the bits of autograd's accumulation) instead of in a
    pass of their own."""

    @staticmethod
    def forward(ctx, x, gamma, beta, film, pre, groups, eps, act):
        xc = x if x.is_contiguous() else x.contiguous()
        y, ws, splits = K.gn_act_fwd(xc, gamma, beta, groups, eps, act, film, pre)
        ctx.save_for_backward(xc, gamma, beta, ws)
        ctx.consts = (film, pre)
        ctx.meta = (groups, eps, act, splits)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dx_skip):
        xc, gamma, beta, ws = ctx.saved_tensors
        groups, eps, act, splits = ctx.meta
        film, pre = ctx.consts
        if dy is None:
            return dx_skip, None, None, None, None, None, None, None
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if dx_skip is not None and not dx_skip.is_contiguous():
            dx_skip = dx_skip.contiguous()
        dx = K.gn_act_bwd(xc, dy, gamma, beta, groups, eps, act, film, ws, splits, pre, add=dx_skip)
        return dx, None, None, None, None, None, None, None


class _BiasAdd2(torch.autograd.Function):
    """(h + bias_c) + other in one pass; the gradient reaches h and other unchanged."""

    @staticmethod
    def forward(ctx, h, bias, other):
        return K.bias_add2(h if h.is_contiguous() else h.contiguous(), bias, other if other.is_contiguous() else other.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, None, g


def fused_glue(x, *consts):
    """True when the HIP glue kernels serve this call: fp32 GPU activations whose spatial size is a multiple of 4 and
    constants (parameters, embedding terms) that carry no gradient; NHMC_FUSED_GN=0 keeps the ATen sequence (A/B runs)."""
    return x.is_cuda and x.dtype == torch.float32 and x[0, 0].numel() % 4 == 0 and os.environ.get('NHMC_FUSED_GN', '1') != '0' \
        and not any(c is not None and c.requires_grad for c in consts)


def group_norm_act(gn, x, act=True, film=None, act_fn=F.silu, pre=None):
    """GroupNorm `gn` of x (+ `pre`, a [C] or [B, C] term added first: the producing convolution's bias / an embedding
    term) (+ FiLM terms `film` = [B, 2C] scale | shift) (+ SiLU) of a score network.

    fp32 GPU tensors with frozen parameters -- the sampler's case -- run the fused HIP kernels; anything else (CPU
    tensors of the fixtures' side, float64 evaluation in the parity tests, parameters that require grad) is plain torch,
    as for any other module of this file -- with `act_fn` the activation in the reference's own form (nn.SiLU in the ADM
    U-Nets, x * sigmoid(x) in the VQ decoder)."""
    if fused_glue(x, gn.weight, gn.bias, film, pre):
        return _GroupNormAct.apply(x, gn.weight, gn.bias, film, pre, gn.num_groups, gn.eps, act)
    if pre is not None:
        x = x + pre.reshape(((1, -1) if pre.dim() == 1 else tuple(pre.shape)) + (1,) * (x.dim() - 2))
    h = F.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
    if film is not None:
        scale, shift = film.reshape(film.shape + (1,) * (x.dim() - 2)).chunk(2, dim=1)
        h = h * (1 + scale) + shift
    return act_fn(h) if act else h


def group_norm_act_fork(gn, x, act=True, act_fn=F.silu):
    """(group_norm_act(gn, x), x) for a block input that also feeds the block's skip path: on the fused kernels the
    second element is an alias of x whose gradient is added inside the GroupNorm's backward kernel."""
    if x.requires_grad and torch.is_grad_enabled() and fused_glue(x, gn.weight, gn.bias):
        return _GroupNormActFork.apply(x, gn.weight, gn.bias, None, None, gn.num_groups, gn.eps, act)
    return group_norm_act(gn, x, act=act, act_fn=act_fn), x


def conv_nobias(conv, x):
    """The convolution without its bias (the fused glue adds it where the output is consumed)."""
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


class _Resample2x(torch.autograd.Function):
    """2x2 average pool / nearest 2x upsample on the block-mean kernels of the SR operator (csrc/data_term.hip: `k_sr`
    with ratio 2 is exactly this pair, H = block mean and H^T = broadcast x scale): each is the other's backward, so
    ATen's slow avg_pool2d_backward (429 us per call at the U-Net's sizes) never runs."""

    @staticmethod
    def forward(ctx, x, up):
        B, C, d = x.shape[0], x.shape[1], x.shape[2]
        ctx.up, ctx.dims = up, (B, C, d)
        xc = x if x.is_contiguous() else x.contiguous()
        if up:
            return K.sr_Ht(xc.reshape(B, -1), 2, C, 2 * d, 1.0).view(B, C, 2 * d, 2 * d)
        return K.sr_H(xc, 2).view(B, C, d // 2, d // 2)

    @staticmethod
    def backward(ctx, g):
        B, C, d = ctx.dims
        gc = g if g.is_contiguous() else g.contiguous()
        if ctx.up:                                          # sum over each 2x2 block = 4 x its mean
            return K.sr_H(gc, 2).view(B, C, d, d).mul_(4.0), None
        return K.sr_Ht(gc.reshape(B, -1), 2, C, d, 0.25).view(B, C, d, d), None


class Resample(nn.Module):
    """Parameter-free 2x nearest upsample / 2x2 average pool (the reference's conv-less up/down)."""

    def __init__(self, up):
        super().__init__()
        self.up = up

    def forward(self, x):
        if fused_glue(x) and x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[0] <= 65535 \
                and (x.shape[2] % 8 == 0 or (self.up and x.shape[2] % 2 == 0)):
            return _Resample2x.apply(x, self.up)
        return F.interpolate(x, scale_factor=2, mode='nearest') if self.up else F.avg_pool2d(x, 2)


class ResBlock(nn.Module):
    takes_emb = True

    def __init__(self, ch, emb_ch, out_ch=None, up=False, down=False):
        super().__init__()
        out_ch = out_ch or ch
        self.in_layers = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(ch, out_ch, 3, padding=1))
        self.resample = up or down
        self.h_upd = self.x_upd = Resample(up) if self.resample else nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, 2 * out_ch))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_ch), nn.SiLU(), nn.Identity(),
                                        nn.Conv2d(out_ch, out_ch, 3, padding=1))
        self.skip_connection = nn.Identity() if out_ch == ch else nn.Conv2d(ch, out_ch, 1)

    def forward(self, x, emb):
        h, x = group_norm_act_fork(self.in_layers[0], x)                 # GroupNorm + SiLU; x goes on to the skip path
        if self.resample:
            h, x = self.h_upd(h), self.x_upd(x)
        conv1, conv2, film = self.in_layers[2], self.out_layers[3], self.emb_layers(emb)
        if fused_glue(h, conv1.bias, conv2.bias, film):
            # the two convolutions run without their broadcast bias passes: conv1's bias enters the next GroupNorm's
            # load, conv2's the residual add
            h = group_norm_act(self.out_layers[0], conv_nobias(conv1, h), film=film, pre=conv1.bias)
            return _BiasAdd2.apply(conv_nobias(conv2, h), conv2.bias, self.skip_connection(x))
        h = group_norm_act(self.out_layers[0], conv1(h), film=film)      # scale-shift norm (FiLM) + SiLU
        return self.skip_connection(x) + conv2(h)


class AttentionBlock(nn.Module):
    takes_emb = False

    def __init__(self, ch, head_ch):
        super().__init__()
        self.heads = ch // head_ch
        self.norm = nn.GroupNorm(32, ch)
        self.qkv = nn.Conv1d(ch, 3 * ch, 1)
        self.proj_out = nn.Conv1d(ch, ch, 1)

    def forward(self, x):
        b, c, hh, ww = x.shape
        x = x.reshape(b, c, -1)
        qkv = self.qkv(group_norm_act(self.norm, x, act=False))
        # "legacy" order: heads are split before q/k/v
        q, k, v = qkv.reshape(b * self.heads, 3 * c // self.heads, -1).split(c // self.heads, dim=1)
        s = 1 / math.sqrt(math.sqrt(c // self.heads))
        w = torch.softmax(torch.einsum('bct,bcs->bts', q * s, k * s), dim=-1)
        h = torch.einsum('bts,bcs->bct', w, v).reshape(b, c, -1)
        return (x + self.proj_out(h)).reshape(b, c, hh, ww)


class Stage(nn.Sequential):
    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if getattr(layer, 'takes_emb', False) else layer(x)
        return x


class UNetModel(nn.Module):
    def __init__(self, image_size=256, in_channels=3, model_channels=128, out_channels=6, num_res_blocks=1,
                 attention_ds=(16,), channel_mult=(1, 1, 2, 2, 4, 4), num_head_channels=64):
        super().__init__()
        self.model_channels = mc = model_channels
        emb = 4 * mc
        self.time_embed = nn.Sequential(nn.Linear(mc, emb), nn.SiLU(), nn.Linear(emb, emb))
        # resident copy of the embedding frequencies (not a checkpoint key): no host->device copy per call, which also
        # keeps the forward capturable into a hipGraph
        self.register_buffer('_freqs', sinusoid_freqs(mc), persistent=False)
        ch = int(channel_mult[0] * mc)
        self.input_blocks = nn.ModuleList([Stage(nn.Conv2d(in_channels, ch, 3, padding=1))])
        skip_chs, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, emb, int(mult * mc))]
                ch = int(mult * mc)
                if ds in attention_ds:
                    layers.append(AttentionBlock(ch, num_head_channels))
                self.input_blocks.append(Stage(*layers))
                skip_chs.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(Stage(ResBlock(ch, emb, ch, down=True)))
                skip_chs.append(ch)
                ds *= 2
        self.middle_block = Stage(ResBlock(ch, emb), AttentionBlock(ch, num_head_channels), ResBlock(ch, emb))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + skip_chs.pop(), emb, int(mc * mult))]
                ch = int(mc * mult)
                if ds in attention_ds:
                    layers.append(AttentionBlock(ch, num_head_channels))
                if level and i == num_res_blocks:
                    layers.append(ResBlock(ch, emb, ch, up=True))
                    ds //= 2
                self.output_blocks.append(Stage(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(ch, out_channels, 3, padding=1))

    def forward(self, x, timesteps, y=None):
        emb = self.time_embed(sinusoid(timesteps, self.model_channels, self._freqs))
        hs, h = [], x
        for blk in self.input_blocks:
            h = blk(h, emb)
            hs.append(h)
        h = self.middle_block(h, emb)
        for blk in self.output_blocks:
            h = blk(torch.cat([h, hs.pop()], dim=1), emb)
        return self.out[2](group_norm_act(self.out[0], h))


def create_model(image_size=256, num_channels=128, num_res_blocks=1, channel_mult='', learn_sigma=True,
                 class_cond=False, use_checkpoint=False, attention_resolutions='16', num_heads=4,
                 num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
                 resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path=''):
    """Keyword-compatible with the reference factory (unet_ffhq.py:25-91) for the options the FFHQ
    config uses; anything that would change the architecture away from it is rejected loudly."""
    if class_cond or use_fp16 or use_new_attention_order or not use_scale_shift_norm or not resblock_updown \
            or num_head_channels <= 0 or dropout:
        raise NotImplementedError('only the FFHQ-config architecture variant is built here')
    mult = _DEFAULT_MULT[image_size] if channel_mult == '' else tuple(int(c) for c in str(channel_mult).split(','))
    att = tuple(image_size // int(r) for r in str(attention_resolutions).split(','))
    model = UNetModel(image_size, 3, num_channels, 6 if learn_sigma else 3, num_res_blocks, att, mult, num_head_channels)
    import os
    if model_path and os.path.exists(model_path):
        model.load_state_dict(torch.load(model_path, map_location='cpu', weights_only=True))
    elif model_path:
        print(f'checkpoint {model_path} not found / randomly initialised')
    return model
