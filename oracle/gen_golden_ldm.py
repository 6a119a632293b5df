"""Generate tests/golden/g12_ldm_16.npz by RUNNING THE REFERENCE's LDM modules and `hmc_latent` (build container only).

Same rules as oracle/gen_golden.py: /root/reference is imported read-only, inert placeholder modules stand in for the
absent off-path packages, only data is written to the repo.

What is captured (all with seeded weights at a reduced width, and the key layout at the FFHQ width):
  * `ldm.modules.diffusionmodules.openaimodel.UNetModel` (openaimodel.py:413) -- state-dict key/shape hashes (reduced
    and configs/config_ffhq_latent.yml:45-65 width) and a forward output;
  * `ldm.modules.diffusionmodules.model.Decoder` (model.py:462) -- the same for the VQ-f4 decoder (:66-80);
  * `make_beta_schedule` (util.py:21-25) -> alphas_cumprod as `register_schedule` builds them (ddpm.py:117-138);
  * the whole `hmc_latent` run (main_sampling_latent.py:623-762) on a model object made of those two reference
    networks.  `LatentDiffusion` itself cannot be instantiated (pytorch_lightning, taming absent), so the object is
    oracle.ldm_ref.OracleLatent: `apply_model` under no_grad as ddpm.py:892 has it, decode = restated taming
    quantiser (not vendored: "parity unpinned" for that step) -> post_quant_conv -> reference Decoder.
"""
import argparse
import contextlib
import io
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, np32, save  # noqa: E402

UNET_SMALL = dict(image_size=16, in_channels=3, out_channels=3, model_channels=32, attention_resolutions=[8, 4, 2],
                  num_res_blocks=2, channel_mult=[1, 2, 3, 4], num_head_channels=32)
UNET_FFHQ = dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, attention_resolutions=[8, 4, 2],
                 num_res_blocks=2, channel_mult=[1, 2, 3, 4], num_head_channels=32)
DEC_SMALL = dict(double_z=False, z_channels=3, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
                 num_res_blocks=2, attn_resolutions=[], dropout=0.0)
DEC_FFHQ = dict(DEC_SMALL, resolution=256, ch=128)
N_EMBED = 4096
HMC = dict(tau=float(os.environ.get('G12_TAU', 0.15)), epsilon=float(os.environ.get('G12_EPS', 0.05)),
           sigma_y=float(os.environ.get('G12_SY', 1.0)), out_gain=float(os.environ.get('G12_GAIN', 0.3)))


def main():
    import_reference()
    om = types.ModuleType('omegaconf')
    om.OmegaConf = None
    sys.modules['omegaconf'] = om
    import main_sampling_latent as msl
    msl.device = torch.device('cpu')
    msl.config = msl.dict2namespace({'data': {'rescaled': True, 'logit_transform': False}})
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.modules.diffusionmodules.model import Decoder
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    from obs_functions.Hfuncs import Inpainting
    from algos.unconditional_latent import Unconditional_Latent
    from oracle.ldm_ref import OracleLatent, vq_straight_through, seeded_state, keys_hash

    with contextlib.redirect_stdout(io.StringIO()):
        unet, dec = UNetModel(**UNET_SMALL).eval(), Decoder(**DEC_SMALL).eval()
        with torch.device('meta'):
            unet_full, dec_full = UNetModel(**UNET_FFHQ), Decoder(**DEC_FFHQ)
    unet.load_state_dict(seeded_state(unet.state_dict(), 1201))
    dsd = seeded_state(dec.state_dict(), 1202)
    dsd['conv_out.weight'] = dsd['conv_out.weight'] * HMC['out_gain']          # decoded images of O(1) amplitude
    dec.load_state_dict(dsd)
    g = torch.Generator().manual_seed(1203)
    pqc = torch.nn.Conv2d(3, 3, 1)
    pqc.load_state_dict(seeded_state(pqc.state_dict(), 1204))
    codebook = torch.rand(N_EMBED, 3, generator=g) * 2 - 1
    for m in (unet, dec, pqc):
        m.requires_grad_(False)

    x = torch.randn(2, 3, 16, 16, generator=g)
    t = torch.tensor([750.0, 250.0])
    z = (torch.rand(2, 3, 16, 16, generator=g) * 2 - 1)
    with torch.no_grad():
        unet_out = unet(x, t)
        dec_out = dec(z)
        zq, idx = vq_straight_through(z, codebook)
        first_stage_out = dec(pqc(zq))

    # schedule as register_schedule computes it from the reference's make_beta_schedule
    betas = make_beta_schedule('linear', 1000, linear_start=0.0015, linear_end=0.0195)
    betas = betas.numpy() if hasattr(betas, 'numpy') else betas
    ac = np.cumprod(1. - betas, axis=0)
    ac_prev = np.append(1., ac[:-1])

    # whole hmc_latent run (fp32, as the reference runs it; its GroupNorm32 / timestep embedding pin fp32 internally, so
    # these classes cannot be evaluated in float64 without editing them)
    model = OracleLatent(unet, dec, pqc, codebook, linear_start=0.0015, linear_end=0.0195)
    assert np.array_equal(model.alphas_cumprod.numpy(), ac.astype(np.float32))
    dim = 64
    r = 3 * torch.randperm(dim * dim, generator=g)[: int(dim * dim * 0.92)].long()
    missing = torch.cat([r, r + 1, r + 2])
    Hf = Inpainting(3, dim, missing, 'cpu')
    x_orig = torch.rand(1, 3, dim, dim, generator=g) * 2 - 1
    sigma_0 = 2 * 0.05
    y_0 = Hf.H(x_orig) + sigma_0 * torch.randn(1, Hf.kept_indices.numel(), generator=g)
    x0 = torch.randn(1, 3, 16, 16, generator=g)
    opt = argparse.Namespace(tau=HMC['tau'], epsilon=HMC['epsilon'], m=1.0, sigma_0=sigma_0, sigma_y=HMC['sigma_y'], algo='hmc_latent', noise='ddpm',
                             image_folder='/tmp/nhmc_golden_scratch')
    algo = Unconditional_Latent(model, Hf, sigma_0)
    rec = dict(p=[], u=[], neg_dH=[])
    real_randn_like, real_rand, real_exp = torch.randn_like, torch.rand, torch.exp

    def randn_like(*a, **k):
        out = real_randn_like(*a, **k)
        rec['p'].append(out.clone())
        return out

    def rand(*a, **k):
        out = real_rand(*a, **k)
        rec['u'].append(float(out.reshape(-1)[0]))
        return out

    def exp(tt, *a, **k):
        if tt.numel() == 1 and tt.dim() == 1:
            rec['neg_dH'].append(float(tt.detach().reshape(-1)[0]))
        return real_exp(tt, *a, **k)

    torch.manual_seed(5678)
    torch.randn_like, torch.rand, torch.exp = randn_like, rand, exp
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = msl.hmc_latent(x0.clone(), 1, [250, 500, 750], [-1, 250, 500], algo, opt, y_0, Hf, x_orig)
    finally:
        torch.randn_like, torch.rand, torch.exp = real_randn_like, real_rand, real_exp

    save('g12_ldm_16.npz',
         unet_keys_small_sha256=np.array(keys_hash(unet.state_dict())),
         unet_keys_ffhq_sha256=np.array(keys_hash(unet_full.state_dict(), 'model.diffusion_model.')),
         unet_n_params_ffhq=np.array(sum(v.numel() for v in unet_full.state_dict().values())),
         dec_keys_small_sha256=np.array(keys_hash(dec.state_dict())),
         dec_keys_ffhq_sha256=np.array(keys_hash(dec_full.state_dict(), 'first_stage_model.decoder.')),
         dec_n_params_ffhq=np.array(sum(v.numel() for v in dec_full.state_dict().values())),
         seeds=np.array([1201, 1202, 1204]), codebook=np32(codebook),
         x=np32(x), t=np32(t), unet_out=np32(unet_out), z=np32(z), dec_out=np32(dec_out), vq_idx=idx.numpy().astype(np.int32),
         vq_out=np32(zq), first_stage_out=np32(first_stage_out),
         alphas_cumprod=ac.astype(np.float32), alphas_cumprod_prev=ac_prev.astype(np.float32),
         hmc_x=np32(x0), hmc_y_0=np32(y_0), hmc_x_orig=np32(x_orig), hmc_missing=np32(missing), hmc_sigma_0=np.array(sigma_0),
         hmc_sigma_y=np.array(HMC['sigma_y']), hmc_tau=np.array(HMC['tau']), hmc_epsilon=np.array(HMC['epsilon']),
         dec_out_gain=np.array(HMC['out_gain']), hmc_out=np32(out),
         hmc_u=np.array(rec['u']), hmc_neg_dH=np.array(rec['neg_dH']), hmc_p=np32(torch.stack(rec['p'])))
    print('hmc_latent iterations', len(rec['u']), 'accepted-ish', sum(u < math.exp(min(0.0, d)) for u, d in zip(rec['u'], rec['neg_dH'])),
          'returned', tuple(out.shape))


if __name__ == '__main__':
    main()
