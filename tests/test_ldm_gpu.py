"""GPU: the latent path on nhmc.ldm's LDM U-Net + VQ first stage (BASELINE configs[4]) -- the codebook kernel, the
networks against the reference-class fixture (G12), the sampler against the latent oracle and the reference's run."""
import types

import numpy as np
import pytest
import torch

from oracle import latent_ref, ldm_ref, operators as oops
from tests.test_ldm_cpu import DEC_SMALL, SEQ, SEQ_NEXT, UNET_SMALL, small_model

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


class F64Product:
    """nhmc.ldm.LatentDiffusion on the GPU with both networks evaluated in float64 (ldm_ref.F64Net); the codebook
    lookup stays the fp32 HIP kernel.  Twin of ldm_ref.OracleLatent.from_product(model, f64=True)."""

    def __init__(self, model, dev):
        fs = model.first_stage_model
        self.unet = ldm_ref.F64Net(model.model.diffusion_model).to(dev)
        self.dec = ldm_ref.F64Net(torch.nn.Sequential(fs.post_quant_conv, fs.decoder)).to(dev)
        self.m = model.to(dev)
        self.alphas_cumprod, self.alphas_cumprod_prev = self.m.alphas_cumprod, self.m.alphas_cumprod_prev

    def apply_model(self, x, t, cond=None):
        with torch.no_grad():
            return self.unet(x, t)

    def differentiable_decode_first_stage(self, z):
        return self.dec(self.m.first_stage_model.quantize(1. / self.m.scale_factor * z))


def test_vq_kernel_takes_the_quantisers_decisions(golden):
    import nhmc.kernels as K
    g = golden('g12_ldm_16.npz')
    dev = torch.device('cuda')
    zq, idx = K.vq_nearest(T(g['z']).to(dev), T(g['codebook']).to(dev))
    assert np.array_equal(idx.cpu().numpy(), g['vq_idx']) and np.array_equal(zq.cpu().numpy(), g['vq_out'])
    gen = torch.Generator().manual_seed(8)
    for D, n_embed, hw in ((3, 8192, 64), (4, 8192, 64), (3, 1000, 24)):          # 1000 codes: ragged last LDS chunk
        z = torch.rand(2, D, hw, hw, generator=gen) * 2 - 1
        cb = torch.randn(n_embed, D, generator=gen) * 0.5
        want_q, want_i = ldm_ref.vq_straight_through(z, cb)
        got_q, got_i = K.vq_nearest(z.to(dev), cb.to(dev))
        # same op order as torch's CPU kernels (sequential norms, FMA-chain inner product): the SAME decisions, not
        # merely near-tie-equivalent ones -- a single flipped code moves the decoded image and its gradient by O(1) locally
        assert torch.equal(got_i.cpu().long(), want_i), int((got_i.cpu().long() != want_i).sum())
        assert torch.equal(got_q.cpu(), want_q)
    with pytest.raises(Exception, match='codebook'):
        K.vq_nearest(torch.zeros(1, 3, 4, 4, device=dev), torch.zeros(8, 4, device=dev))
    with pytest.raises(Exception, match='shape'):
        K.vq_nearest(torch.zeros(1, 5, 4, 4, device=dev), torch.zeros(8, 5, device=dev))


def test_vq_kernel_on_non_finite_latents_answers_as_argmin_does_and_stays_in_bounds():
    """a NaN / inf latent pixel (a divergent trajectory decoded through the first stage) must not send the
    codebook gather out of bounds.  torch.argmin (the quantiser's op, taming VectorQuantizer2.forward) counts a NaN distance
    as the minimum and returns the first one, and returns 0 for an all-inf row; the proposal then carries NaN and the
    Metropolis test rejects it."""
    import nhmc.kernels as K
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(9)
    for D in (3, 4):
        z = torch.rand(1, D, 16, 16, generator=gen) * 2 - 1
        cb = torch.randn(8192, D, generator=gen) * 0.5
        z[0, 0, 0, 0] = float('nan')                     # every distance NaN -> index 0
        z[0, 1, 0, 1] = float('inf')                     # NaN where the inner product is +inf, +inf elsewhere -> first NaN
        z[0, 2, 0, 2] = float('-inf')
        z[0, :, 0, 3] = float('inf')
        z[0, :, 0, 4] = 3.0e38                           # finite, but the squared norm overflows: distances +inf or NaN
        want_q, want_i = ldm_ref.vq_straight_through(z, cb)
        got_q, got_i = K.vq_nearest(z.to(dev), cb.to(dev))
        torch.cuda.synchronize()
        assert int(got_i.min()) >= 0 and int(got_i.max()) < 8192
        assert torch.equal(got_i.cpu().long(), want_i), (got_i.cpu().long() != want_i).nonzero()[:5]
        assert int(want_i[0, 0, 0]) == 0                 # all NaN: argmin answers 0
        assert torch.equal(torch.isnan(got_q.cpu()), torch.isnan(want_q))
        fin = torch.isfinite(want_q)
        assert torch.equal(got_q.cpu()[fin], want_q[fin])


def test_networks_on_the_gpu_match_the_reference_classes(golden):
    g = golden('g12_ldm_16.npz')
    dev = torch.device('cuda')
    m = small_model(g).to(dev)
    with torch.no_grad():
        out = m.apply_model(T(g['x']).to(dev), T(g['t']).to(dev), None)
        img = m.first_stage_model.decoder(T(g['z']).to(dev))
        full = m.decode_first_stage(T(g['z']).to(dev))
    assert rel(out, T(g['unet_out'])) < 1e-4 and rel(img, T(g['dec_out'])) < 1e-4
    assert rel(full, T(g['first_stage_out'])) < 1e-4
    # straight-through gradient of the first stage reaches the latent
    z = T(g['z']).to(dev).requires_grad_(True)
    (gz,) = torch.autograd.grad(m.differentiable_decode_first_stage(z).sum(), z)
    twin = ldm_ref.OracleLatent.from_product(m)
    zc = T(g['z']).clone().requires_grad_(True)
    (gc,) = torch.autograd.grad(twin.differentiable_decode_first_stage(zc).sum(), zc)
    assert rel(gz, gc) < 1e-4


def _engine(model, op, dev):
    from nhmc import plugin, sampler
    algo = plugin.HMCLatent(model, op, 0.1)
    table = torch.cat([model.alphas_cumprod_prev[0:1], model.alphas_cumprod])
    return sampler.LeapfrogEngine(algo.score, op, None, SEQ, SEQ_NEXT, dev, alpha_table=table,
                                  image_map=model.differentiable_decode_first_stage), algo


def test_first_trajectory_of_the_reference_ldm_run(golden):
    """The reference's `hmc_latent` on its own UNetModel / Decoder classes (G12), first trajectory: positions, decode
    and loss against the oracle, and the energy change against the -dH the reference itself recorded.  The networks are
    evaluated in float64 on both sides: through the codebook lookup an fp32 convolution-rounding difference between CPU
    and GPU can flip one code and move the decoded image by O(1) locally (observed: the fp32 form of this test passed or
    failed depending on which solver MIOpen had picked earlier in the process), which says nothing about the sampler."""
    from nhmc import operators, sampler
    g = golden('g12_ldm_16.npz')
    dev = torch.device('cuda')
    base = small_model(g)
    L = max(1, int(np.floor(float(g['hmc_tau']) / float(g['hmc_epsilon']))))
    want = latent_ref.trajectory_latent(T(g['hmc_x']), T(g['hmc_p'][0]), SEQ, SEQ_NEXT, ldm_ref.OracleLatent.from_product(base, f64=True),
                                        oops.InpaintRef(3, 64, T(g['hmc_missing']).long()), T(g['hmc_y_0']),
                                        sigma_y=float(g['hmc_sigma_y']), eps=float(g['hmc_epsilon']), m=1.0, L=L)
    op = operators.Inpainting(3, 64, T(g['hmc_missing']).long(), dev)
    eng, _ = _engine(F64Product(base, dev), op, dev)
    st = sampler.ChainState(1, float(g['hmc_tau']), float(g['hmc_epsilon']), dev)
    st['eps_eff'].fill_(float(g['hmc_epsilon']))
    st['sigma_y'].fill_(float(g['hmc_sigma_y']))
    got = sampler.run_trajectory(eng, T(g['hmc_x']).to(dev), T(g['hmc_p'][0]).to(dev).clone(), T(g['hmc_y_0']).to(dev), st, 1.0, L)
    assert rel(got['x_prop'], want['x']) < 1e-4 and rel(got['xt'], want['xt']) < 1e-4 and rel(got['loss'], want['loss']) < 1e-4
    assert abs(float((want['H1'] - want['H0'])[0]) + float(g['hmc_neg_dH'][0])) < 0.02       # oracle == the reference's own -dH
    assert abs(float((got['H1'] - got['H0'])[0]) + float(g['hmc_neg_dH'][0])) < 0.02        # and so is the GPU's


def test_latent_loop_on_the_ldm_model_takes_the_oracles_decisions(golden):
    """Whole 70-epoch `hmc_latent` loop on the tape the reference drew (G12), networks in float64 on both sides:
    same accept decisions as the oracle, same returned latents; and the oracle in fp32 is pinned to the reference's
    own run on CPU (tests/test_ldm_cpu.py)."""
    from nhmc import operators, plugin, sampler
    g = golden('g12_ldm_16.npz')
    dev = torch.device('cuda')
    base = small_model(g)
    ref_op = oops.InpaintRef(3, 64, T(g['hmc_missing']).long())
    op = operators.Inpainting(3, 64, T(g['hmc_missing']).long(), dev)
    P, U = T(g['hmc_p']), T(g['hmc_u']).float()
    kw = dict(sigma_y=float(g['hmc_sigma_y']), tau=float(g['hmc_tau']), epsilon=float(g['hmc_epsilon']), m=1.0,
              sigma_0=float(g['hmc_sigma_0']))

    class Tape:                                                   # the oracle draws through torch's global functions
        def __init__(self):
            self.ip = self.iu = 0

        def randn_like(self, x, **k):
            self.ip += 1
            return P[self.ip - 1].clone()

        def rand(self, *a, **k):
            self.iu += 1
            return U[self.iu - 1].reshape(1).clone()

    tape, trace = Tape(), {}
    real = torch.randn_like, torch.rand
    torch.randn_like, torch.rand = tape.randn_like, tape.rand
    try:
        want = latent_ref.hmc_latent_reference(T(g['hmc_x']), SEQ, SEQ_NEXT, ldm_ref.OracleLatent.from_product(base, f64=True),
                                               ref_op, T(g['hmc_y_0']), T(g['hmc_x_orig']), trace=trace, **kw)
    finally:
        torch.randn_like, torch.rand = real
    algo = plugin.HMCLatent(F64Product(base, dev), op, kw['sigma_0'])
    opt = types.SimpleNamespace(**kw)
    res = sampler.hmc_latent_chains(T(g['hmc_x']).to(dev), SEQ, SEQ_NEXT, algo, opt, T(g['hmc_y_0']).to(dev), op,
                                    T(g['hmc_x_orig']).to(dev),
                                    noise=sampler.TapeNoise(lambda it: P[it], lambda it: U[it].reshape(1)), collect_trace=True)
    got_acc = [bool(r['accept'][0]) for r in res.trace]
    for it, (a, b) in enumerate(zip(trace['accept'], got_acc)):
        margin = abs(float(U[it]) - min(1.0, float(np.exp(-trace['dH'][it]))))
        assert a == b or margin < 1e-3, (it, a, b, margin)
    assert [float(r['sigma_y'][0]) for r in res.trace] == trace['sigma_y']
    assert res.samples[0].shape == want.shape and rel(res.samples[0], want) < 1e-4
    # and the reference's own (fp32) decisions: identical wherever its accept probability is not within the band
    ref_acc = [bool(u < np.exp(min(0.0, d))) for u, d in zip(g['hmc_u'], g['hmc_neg_dH'])]
    n_prefix = next((i for i, (a, b) in enumerate(zip(ref_acc, got_acc)) if a != b), len(ref_acc))
    assert n_prefix >= 10, n_prefix        # fp32-vs-fp64 network noise may separate the runs later, not at the start


@pytest.mark.parametrize('C,B', [(3, 16), (4, 2)])
def test_configs4_size_trajectory_against_the_oracle(C, B):
    """BASELINE configs[4] geometry: [B, C, 64, 64] latents -> 256x256 image through the VQ first stage (8192 codes),
    inpaint_random at 256x256, 16 chains per GPU; networks at a reduced width so the CPU oracle finishes in seconds
    (C = 3 is the reference config, C = 4 is BASELINE.json's wording)."""
    from nhmc import ldm, operators, sampler
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(40 + C)
    unet_cfg = dict(UNET_SMALL, image_size=64, in_channels=C, out_channels=C)
    fs_cfg = dict(embed_dim=C, n_embed=8192, ddconfig=dict(DEC_SMALL, resolution=256, z_channels=C))
    model = ldm.LatentDiffusion(unet_config=unet_cfg, first_stage_config=fs_cfg, linear_start=0.0015, linear_end=0.0195)
    model.load_state_dict({**model.state_dict(), **ldm_ref.seeded_state(
        {k: v for k, v in model.state_dict().items() if not k.startswith('alphas')}, 4400 + C)})
    with torch.no_grad():
        model.first_stage_model.decoder.conv_out.weight.mul_(0.3)
        model.first_stage_model.quantize.embedding.weight.copy_(torch.rand(8192, C, generator=gen) * 2 - 1)
    model = model.eval().requires_grad_(False)
    missing = oops.random_inpaint_missing(256, generator=gen)
    ref_op, op = oops.InpaintRef(3, 256, missing), operators.Inpainting(3, 256, missing, dev)
    assert op.M == 15729
    x = torch.randn(B, C, 64, 64, generator=gen)
    p = torch.randn(B, C, 64, 64, generator=gen)
    y = ref_op.H(torch.rand(B, 3, 256, 256, generator=gen) * 2 - 1) + 0.1 * torch.randn(B, ref_op.M, generator=gen)
    L = 2
    sel = sorted({0, B // 2, B - 1})                       # chains are independent: the fp64 CPU oracle runs three of them
    want = latent_ref.trajectory_latent(x[sel], p[sel].clone(), SEQ, SEQ_NEXT, ldm_ref.OracleLatent.from_product(model, f64=True),
                                        ref_op, y[sel], sigma_y=1.0, eps=0.05, m=1.0, L=L)
    eng, _ = _engine(F64Product(model, dev), op, dev)
    eng.chunk = 8
    st = sampler.ChainState(B, 1.0, 0.05, dev)
    st['eps_eff'].fill_(0.05)
    st['sigma_y'].fill_(1.0)
    got = sampler.run_trajectory(eng, x.to(dev), p.to(dev).clone(), y.to(dev), st, 1.0, L)
    assert got['xt'].shape == (B, C, 64, 64) and bool(torch.isfinite(got['H1']).all())
    assert rel(got['x_prop'][sel], want['x']) < 1e-4 and rel(got['p'][sel], want['p']) < 1e-4
    assert rel(got['xt'][sel], want['xt']) < 1e-4 and rel(got['loss'][sel], want['loss']) < 1e-4
    assert float((got['H1'].cpu()[sel] - want['H1']).abs().max()) <= 1e-4 * float(want['H1'].abs().max())


def test_ffhq_width_latent_model_runs_one_trajectory():
    """The configs/config_ffhq_latent.yml architecture at full width (random init: the checkpoints are fetch-only),
    2 chains: a trajectory completes with finite energies, and eps = 0 leaves the position where it was."""
    from nhmc import ldm, operators, sampler
    dev = torch.device('cuda')
    torch.manual_seed(3)
    model = ldm.create_latent_model(ckpt=None, quiet=True).to(dev)
    gen = torch.Generator().manual_seed(5)
    op = operators.Inpainting(3, 256, oops.random_inpaint_missing(256, generator=gen), dev)
    eng, _ = _engine(model, op, dev)
    x = torch.randn(2, 3, 64, 64, generator=gen).to(dev)
    p = torch.randn(2, 3, 64, 64, generator=gen).to(dev)
    y = torch.randn(2, op.M, generator=gen).to(dev)
    st = sampler.ChainState(2, 1.0, 0.05, dev)
    st['sigma_y'].fill_(1.0)
    got = sampler.run_trajectory(eng, x, p.clone(), y, st, 1.0, 1)                   # eps_eff = 0: frozen chains
    assert torch.equal(got['x_prop'], x) and torch.equal(got['p'], p)
    # same position twice -> same energy, up to the convolution library answering its first call with another solver
    # (a last-bit difference there can flip one of the 2 x 4096 codebook decisions: allow that much)
    assert bool(torch.isfinite(got['H0']).all()) and float((got['H0'] - got['H1']).abs().max()) <= 5e-3 * float(got['H0'].abs().max())
    st['eps_eff'].fill_(0.05)
    got = sampler.run_trajectory(eng, x, p.clone(), y, st, 1.0, 1)
    assert bool(torch.isfinite(got['H1']).all()) and bool(torch.isfinite(got['xt']).all()) and not torch.equal(got['x_prop'], x)
    img = model.decode_first_stage(got['xt'])
    assert img.shape == (2, 3, 256, 256)


def test_fp32_product_path_at_configs4_width_against_float64():
    """the path `bench.py by_deg.hmc_latent` times -- configs/config_ffhq_latent.yml at full width in
    fp32: LDM U-Net (fused GroupNorm glue, MIOpen convolutions), DDIM mixes, `k_vq_nearest`, VQ-f4 decoder, inpainting
    data term -- against the SAME modules evaluated in float64 on the GPU (F64Product).  Continuous quantities (U-Net
    output, the clipped decode that enters the quantiser) must agree to 1e-4.  The codebook lookup is discontinuous: a
    latent pixel whose two nearest codes are closer than the fp32 network noise flips; the fraction of flipped codes is
    reported and bounded, and loss / energy difference must sit inside the band that fraction explains (each flipped code
    moves a 4 x 4 image patch by O(1): the loss moves by about its share of the pixels)."""
    import nhmc.kernels as K
    from nhmc import ldm, operators, sampler
    dev = torch.device('cuda')
    torch.manual_seed(3)
    model = ldm.create_latent_model(ckpt=None, quiet=True).to(dev)
    f64 = F64Product(model, dev)
    gen = torch.Generator().manual_seed(6)
    op = operators.Inpainting(3, 256, oops.random_inpaint_missing(256, generator=gen), dev)
    B = 4
    x = torch.randn(B, model.channels, 64, 64, generator=gen).to(dev)
    p = torch.randn(B, model.channels, 64, 64, generator=gen).to(dev)
    y = (torch.rand(B, op.M, generator=gen) * 2 - 1).to(dev)
    t = torch.full((B,), 500.0, device=dev)
    e32, e64 = model.apply_model(x, t), f64.apply_model(x, t)
    err_unet = rel(e32, e64)
    eng32, _ = _engine(model, op, dev)
    eng64, _ = _engine(f64, op, dev)
    z32, z64 = eng32.decode(x), eng64.decode(x)                      # 3 x (U-Net + DDIM mix) + final clip: what the quantiser sees
    err_z = rel(z32, z64)
    cb = model.first_stage_model.quantize.embedding.weight.detach().contiguous()
    _, i32 = K.vq_nearest(z32.contiguous(), cb)
    _, i64 = K.vq_nearest(z64.contiguous(), cb)
    flipped = float((i32 != i64).float().mean())
    st = sampler.ChainState(B, 1.0, 0.1, dev)
    st['eps_eff'].fill_(0.1)
    st['sigma_y'].fill_(0.5)
    g32 = sampler.run_trajectory(eng32, x, p.clone(), y, st, 1.0, 2)
    r32 = {k: g32[k].clone() for k in ('x_prop', 'xt', 'loss', 'H0', 'H1')}
    g64 = sampler.run_trajectory(eng64, x, p.clone(), y, st, 1.0, 2)
    err_loss = rel(r32['loss'], g64['loss'])
    dH32, dH64 = (r32['H1'] - r32['H0']).cpu(), (g64['H1'] - g64['H0']).cpu()
    k = 1 / (2 * 0.5 ** 2)
    band = float(k * g64['loss'].max()) * (4 * flipped + 1e-5) + 2e-5 * float(g64['H1'].abs().max())
    print(f'configs[4] width, fp32 vs float64 on the GPU: U-Net output {err_unet:.2e}, clipped decode {err_z:.2e}, flipped VQ codes '
          f'{flipped:.2e} of {i32.numel()}, loss {err_loss:.2e}, |dH32 - dH64| max {float((dH32 - dH64).abs().max()):.3f} (band {band:.3f}), '
          f'positions {rel(r32["x_prop"], g64["x_prop"]):.2e}')
    assert err_unet < 1e-4 and err_z < 1e-4
    assert flipped < 5e-3
    assert err_loss < 4 * flipped + 1e-4
    assert float((dH32 - dH64).abs().max()) < band


def test_latent_cli_runs_the_reference_command_line(tmp_path, monkeypatch, capsys):
    """`main_sampling_latent.py`-compatible flags end to end (main_sampling_latent.py:791-918) on a small latent config
    written in the reference's yaml layout (configs/config_ffhq_latent.yml: target / params nesting)."""
    import yaml
    from nhmc import cli
    cfg = {'data': {'dataset': 'tiny', 'image_size': 64, 'channels': 3, 'rescaled': True},
           'model_type': 'ffhq_latent',
           'model': {'target': 'ldm.models.diffusion.ddpm.LatentDiffusion',
                     'params': {'linear_start': 0.0015, 'linear_end': 0.0195, 'timesteps': 1000, 'image_size': 16, 'channels': 3,
                                'first_stage_key': 'image', 'cond_stage_config': '__is_unconditional__',
                                'unet_config': {'target': 'ldm.modules.diffusionmodules.openaimodel.UNetModel', 'params': UNET_SMALL},
                                'first_stage_config': {'target': 'ldm.models.autoencoder.VQModelInterface',
                                                       'params': {'embed_dim': 3, 'n_embed': 256, 'ckpt_path': 'models/first_stage_models/vq-f4/model.ckpt',
                                                                  'ddconfig': DEC_SMALL, 'lossconfig': {'target': 'torch.nn.Identity'}}}}}}
    (tmp_path / 'configs').mkdir()
    (tmp_path / 'configs' / 'config_tiny_latent.yml').write_text(yaml.safe_dump(cfg))
    monkeypatch.chdir(tmp_path)
    opt = cli.get_parser(latent=True).parse_known_args(['--deg', 'sr4', '--sigma_0', '0.05'])[0]
    assert opt.epsilon == 0.1 and opt.sigma_y == 0.5 and not hasattr(opt, 'annealed_temp')      # the latent entry's defaults
    table = cli.main_latent(['--dataset', 'tiny', '--algo', 'hmc_latent', '--timesteps', '3', '--deg', 'inpaint_random', '--sigma_0', '0.05',
                             '-i', str(tmp_path / 'out'), '--tau', '0.1', '--epsilon', '0.1', '--sigma_y', '1.0', '--synthetic', '2',
                             '--chains', '2', '--philox'])
    assert table.shape == (2, 3) and table[:, 0].tolist() == [0.0, 1.0]
    assert 'Total Average PSNR' in capsys.readouterr().out
    with pytest.raises(NotImplementedError):
        cli.main_latent(['--dataset', 'tiny', '--algo', 'hmc', '--deg', 'sr4', '--sigma_0', '0.05'])
    with pytest.raises(NotImplementedError):
        cli.main(['--dataset', 'tiny', '--algo', 'hmc_latent', '--deg', 'sr4', '--sigma_0', '0.05'])


@pytest.mark.parametrize('deg', ['sr4', 'color', 'cs4', 'deblur_aniso'])
def test_latent_trajectory_with_the_other_operators_on_the_decoded_image(golden, deg):
    """`hmc_latent` applies H to the DECODED image (main_sampling_latent.py:683-684: loss = |y - H(decode(z))|^2), whatever
    the degradation; the engine's latent branch takes the operator's data-term gradient on the image and pulls it back through
    the decoder.  One trajectory on the G12 networks (float64 on both sides, as above) for operators other than inpainting,
    against the latent oracle with the matching reference-order operator."""
    from nhmc import operators, sampler
    g = golden('g12_ldm_16.npz')
    dev = torch.device('cuda')
    base = small_model(g)
    dim, B, L = 64, 2, 2
    gen = torch.Generator().manual_seed(len(deg) * 131)
    if deg == 'sr4':
        ref_op, op = oops.BlockMeanRef(3, dim, 4), operators.SuperResolution(3, dim, 4, dev)
    elif deg == 'color':
        ref_op, op = oops.ColorRef(dim), operators.Colorization(dim, dev)
    elif deg == 'cs4':
        perm = torch.randperm(dim * dim, generator=gen)
        ref_op, op = oops.WalshHadamardRef(3, dim, 4, perm), operators.WalshHadamardCS(3, dim, 4, perm, dev)
    else:
        k1, k2 = oops.gaussian_taps(20.0), oops.gaussian_taps(1.0)
        ref_op = oops.SpectralBlurRef.from_kernels(k1 / k1.sum(), k2 / k2.sum(), 3, dim)
        op = operators.Deblurring2D.from_factors(ref_op.U1, ref_op.U2, ref_op.V1, ref_op.V2, ref_op.D, dev)
    x = torch.randn(B, 3, 16, 16, generator=gen)
    p = torch.randn(B, 3, 16, 16, generator=gen)
    y = ref_op.H(torch.rand(B, 3, dim, dim, generator=gen) * 2 - 1)
    y = y + 0.1 * torch.randn(y.shape, generator=gen)
    want = latent_ref.trajectory_latent(x, p.clone(), SEQ, SEQ_NEXT, ldm_ref.OracleLatent.from_product(base, f64=True), ref_op, y,
                                        sigma_y=1.0, eps=0.05, m=1.0, L=L)
    eng, _ = _engine(F64Product(base, dev), op, dev)
    st = sampler.ChainState(B, 1.0, 0.05, dev)
    st['eps_eff'].fill_(0.05)
    st['sigma_y'].fill_(1.0)
    got = sampler.run_trajectory(eng, x.to(dev), p.to(dev).clone(), y.to(dev), st, 1.0, L)
    assert rel(got['x_prop'], want['x']) < 1e-4 and rel(got['p'], want['p']) < 1e-4
    assert rel(got['xt'], want['xt']) < 1e-4 and rel(got['loss'], want['loss']) < 1e-4
    assert float((got['H1'].cpu() - want['H1']).abs().max()) <= 1e-4 * float(want['H1'].abs().max())
