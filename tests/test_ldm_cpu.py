"""CPU: nhmc.ldm's networks against the reference's LDM classes (G12, oracle/gen_golden_ldm.py) and the latent oracle
against the reference's whole `hmc_latent` run on those classes."""
import numpy as np
import pytest
import torch

from oracle import latent_ref, ldm_ref, operators as oops

T = torch.from_numpy
UNET_SMALL = dict(image_size=16, in_channels=3, out_channels=3, model_channels=32, attention_resolutions=[8, 4, 2],
                  num_res_blocks=2, channel_mult=[1, 2, 3, 4], num_head_channels=32)
DEC_SMALL = dict(double_z=False, z_channels=3, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
                 num_res_blocks=2, attn_resolutions=[], dropout=0.0)
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]


def small_model(g):
    """nhmc.ldm.LatentDiffusion at G12's reduced width with G12's seeded weights."""
    from nhmc import ldm
    m = ldm.LatentDiffusion(unet_config=UNET_SMALL, first_stage_config=dict(embed_dim=3, n_embed=g['codebook'].shape[0], ddconfig=DEC_SMALL),
                            linear_start=0.0015, linear_end=0.0195)
    su, sd_, sp = (int(s) for s in g['seeds'])
    unet, fs = m.model.diffusion_model, m.first_stage_model
    unet.load_state_dict(ldm_ref.seeded_state(unet.state_dict(), su))
    dsd = ldm_ref.seeded_state(fs.decoder.state_dict(), sd_)
    dsd['conv_out.weight'] = dsd['conv_out.weight'] * float(g['dec_out_gain'])
    fs.decoder.load_state_dict(dsd)
    fs.post_quant_conv.load_state_dict(ldm_ref.seeded_state(fs.post_quant_conv.state_dict(), sp))
    fs.quantize.embedding.weight.data.copy_(T(g['codebook']))
    return m.eval().requires_grad_(False)


def test_unet_and_decoder_match_the_reference_classes(golden):
    g = golden('g12_ldm_16.npz')
    m = small_model(g)
    unet, dec = m.model.diffusion_model, m.first_stage_model.decoder
    assert ldm_ref.keys_hash(unet.state_dict()) == str(g['unet_keys_small_sha256'])
    assert ldm_ref.keys_hash(dec.state_dict()) == str(g['dec_keys_small_sha256'])
    with torch.no_grad():
        out = unet(T(g['x']), T(g['t']))
        img = dec(T(g['z']))
    assert out.shape == (2, 3, 16, 16) and img.shape == (2, 3, 64, 64)
    assert float((out - T(g['unet_out'])).abs().max()) <= 1e-5 * float(T(g['unet_out']).abs().max())
    assert float((img - T(g['dec_out'])).abs().max()) <= 1e-5 * float(T(g['dec_out']).abs().max())
    # the model object applies them the way LatentDiffusion does: score without gradient, schedule buffers
    x = T(g['x']).clone().requires_grad_(True)
    assert not m.apply_model(x, T(g['t']), None).requires_grad
    assert np.array_equal(m.alphas_cumprod.numpy(), g['alphas_cumprod'])
    assert np.array_equal(m.alphas_cumprod_prev.numpy(), g['alphas_cumprod_prev'])


def test_checkpoint_layout_at_ffhq_width(golden):
    """Key names and shapes of `models/ldm/model.ckpt` for the parts the path uses, from the reference classes at the
    configs/config_ffhq_latent.yml widths."""
    from nhmc import ldm
    g = golden('g12_ldm_16.npz')
    with torch.device('meta'):
        m = ldm.LatentDiffusion(**ldm.FFHQ_LDM)
    sd = m.state_dict()
    unet = {k: v for k, v in sd.items() if k.startswith('model.diffusion_model.')}
    dec = {k: v for k, v in sd.items() if k.startswith('first_stage_model.decoder.')}
    assert ldm_ref.keys_hash(unet) == str(g['unet_keys_ffhq_sha256'])
    assert ldm_ref.keys_hash(dec) == str(g['dec_keys_ffhq_sha256'])
    assert sum(v.numel() for v in unet.values()) == int(g['unet_n_params_ffhq'])
    assert sum(v.numel() for v in dec.values()) == int(g['dec_n_params_ffhq'])
    assert sd['first_stage_model.quantize.embedding.weight'].shape == (8192, 3)
    assert sd['first_stage_model.post_quant_conv.weight'].shape == (3, 3, 1, 1)
    with pytest.raises(NotImplementedError):
        ldm.LDMUNet(**dict(ldm.FFHQ_LDM_UNET, use_scale_shift_norm=True))


def test_oracle_quantiser_and_first_stage_reproduce_the_fixture(golden):
    g = golden('g12_ldm_16.npz')
    zq, idx = ldm_ref.vq_straight_through(T(g['z']), T(g['codebook']))
    assert np.array_equal(idx.numpy().astype(np.int32), g['vq_idx']) and np.array_equal(zq.numpy(), g['vq_out'])
    twin = ldm_ref.OracleLatent.from_product(small_model(g))
    with torch.no_grad():
        img = twin.differentiable_decode_first_stage(T(g['z']))
    assert float((img - T(g['first_stage_out'])).abs().max()) <= 1e-5 * float(T(g['first_stage_out']).abs().max())
    # straight-through: the quantiser's VJP is the identity
    z = T(g['z']).clone().requires_grad_(True)
    zq, _ = ldm_ref.vq_straight_through(z, T(g['codebook']))
    (gz,) = torch.autograd.grad(zq, z, torch.ones_like(zq))
    assert torch.equal(gz, torch.ones_like(gz))


def test_latent_oracle_reproduces_the_reference_run_on_the_ldm_classes(golden):
    """oracle.latent_ref.hmc_latent_reference on the CPU twin of nhmc.ldm's model == the reference's `hmc_latent` on its
    own UNetModel / Decoder classes: same accept decisions, same returned latents."""
    g = golden('g12_ldm_16.npz')
    twin = ldm_ref.OracleLatent.from_product(small_model(g))
    ref_op = oops.InpaintRef(3, 64, T(g['hmc_missing']).long())
    torch.manual_seed(5678)
    trace = {}
    out = latent_ref.hmc_latent_reference(T(g['hmc_x']), SEQ, SEQ_NEXT, twin, ref_op, T(g['hmc_y_0']), T(g['hmc_x_orig']),
                                          sigma_y=float(g['hmc_sigma_y']), tau=float(g['hmc_tau']), epsilon=float(g['hmc_epsilon']),
                                          m=1.0, sigma_0=float(g['hmc_sigma_0']), trace=trace)
    want_acc = [bool(u < np.exp(min(0.0, d))) for u, d in zip(g['hmc_u'], g['hmc_neg_dH'])]
    assert trace['accept'] == want_acc
    assert np.allclose(-np.array(trace['dH']), g['hmc_neg_dH'], rtol=1e-3, atol=2e-2)
    assert out.shape == g['hmc_out'].shape
    assert float((out - T(g['hmc_out'])).abs().max()) <= 1e-4 * float(T(g['hmc_out']).abs().max())
