"""CPU: host-side logic of the product (schedule tables, operator index maps and factor data,
U-Net architecture, reference-surface signatures) and the oracle's own building blocks."""
import hashlib
import inspect

import numpy as np
import pytest
import torch

from oracle import operators as oops, philox_ref, schedule as osched

T = torch.from_numpy


# ---- schedule ---------------------------------------------------------------------------------
def test_schedule_matches_reference_goldens(golden):
    from nhmc import schedule
    g = golden('g1_schedule.npz')
    betas = schedule.get_beta_schedule('linear', beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)
    assert np.array_equal(betas, g['betas64'])
    b = T(betas).float()
    assert np.array_equal(schedule.alpha_bar_table(b).numpy(), g['table'])
    at = schedule.compute_alpha(b, torch.tensor([750, 500, 250, -1]))
    assert at.shape == (4, 1, 1, 1) and np.array_equal(at.reshape(-1).numpy(), g['at_750_500_250_m1'])
    assert schedule.timestep_ladder(1000, 3) == ([250, 500, 750], [-1, 250, 500])
    assert schedule.timestep_ladder(1000, 10)[0][0] == 90          # skip = 1000 // 11 (main_sampling.py:469)


# ---- operators: host-built constants ------------------------------------------------------------
@pytest.mark.parametrize('dim', [16, 32, 64])
def test_inpainting_index_maps(dim):
    from nhmc import operators
    missing = oops.random_inpaint_missing(dim, generator=torch.Generator().manual_seed(dim))
    ref = oops.InpaintRef(3, dim, missing)
    op = operators.Inpainting(3, dim, missing, 'cpu')
    assert op.M == ref.M and torch.equal(op.kept_indices, ref.kept)
    hw = dim * dim
    # kept_chw / slot are consistent inverses and point at the element the reference's HWC gather reads
    x = torch.randn(2, 3, dim, dim)
    flat = x.reshape(2, -1)
    assert torch.equal(flat[:, op.kept_chw.long()], ref.H(x))
    k = torch.arange(op.M, dtype=torch.int32)
    assert torch.equal(op.slot[op.kept_chw.long()], k)
    assert int((op.slot >= 0).sum()) == op.M and int((op.slot == -1).sum()) == 3 * hw - op.M


def test_inpainting_ffhq_mask_has_the_survey_size():
    from nhmc import operators
    op = operators.build_operator('inpaint_random', 3, 256, 'cpu', generator=torch.Generator().manual_seed(5678))
    assert op.M == 15729                                               # 3 * (65536 - int(65536 * 0.92))


@pytest.mark.parametrize('dim', [32, 64])
def test_spectral_operator_construction(golden, dim):
    """Band matrices (8-of-9 taps), truncation and the multiplier-map layout reproduce the reference's
    exported operator data; the product's own (stable-sort) instance differs only where the reference's
    unstable sort breaks ties among equal singular-value products."""
    from nhmc import operators
    g = golden(f'g2_ops_{dim}.npz')
    op = operators.Deblurring2D(operators.gaussian_taps(1.0), operators.gaussian_taps(20.0), 3, dim, 'cpu')
    assert op.factors.shape == (8, dim, dim) and op.Dmap.shape == (3, dim, dim)
    assert torch.equal(op.factors[4], op.factors[0].t()) and torch.equal(op.factors[7], op.factors[3].t())
    Hs = operators._band_matrix(operators.gaussian_taps(20.0), dim)
    assert torch.equal(Hs, oops.band_matrix(oops.gaussian_taps(20.0), dim)) and int((Hs[dim // 2] != 0).sum()) == 8
    # same multiset of multipliers per channel as the reference's export
    for c in range(3):
        assert np.allclose(np.sort(op.Dmap[c].reshape(-1).numpy()), np.sort(g['aniso_D'][c].reshape(-1)), atol=1e-6)
    # U1 D V1^T style reconstruction: factors are orthogonal
    eye = torch.eye(dim)
    for i in range(4):
        assert torch.allclose(op.factors[i] @ op.factors[i].t(), eye, atol=1e-5)


def test_build_operator_covers_the_hot_path_degradations():
    from nhmc import operators
    assert isinstance(operators.build_operator('sr4', 3, 64, 'cpu'), operators.SuperResolution)
    assert operators.build_operator('sr16', 3, 64, 'cpu').M == 3 * 16
    with pytest.raises(NotImplementedError):
        operators.build_operator('deblur_nonlinear', 3, 64, 'cpu')


def test_every_operator_offers_the_engine_surface():
    """data_term (loss + gradient) and fused_last_vjp (data term handed to the last DDIM-step VJP) for every
    degradation the CLI accepts; operators whose data term works on the clipped decode say so."""
    import inspect
    from nhmc import operators
    for deg in ['sr4', 'sr16', 'sr_bicubic2', 'inpaint_random', 'inpaint_box', 'deblur_aniso', 'deblur_gauss', 'color', 'cs4']:
        op = operators.build_operator(deg, 3, 64, torch.device('cpu'))
        assert callable(op.data_term) and callable(op.fused_last_vjp), deg
        params = inspect.signature(op.fused_last_vjp).parameters
        assert list(params)[:5] == ['xt_in', 'e', 'at', 'at_next', 'y'] and 'g_e_out' in params, deg
        assert ('xt_next' in params) == bool(getattr(op, 'fused_wants_decode', False)), deg
    assert not hasattr(operators.build_operator('sr32', 3, 64, 'cpu'), 'fused_last_vjp')


# ---- reference surfaces ---------------------------------------------------------------------------
def test_plugin_and_sampler_keep_the_reference_signatures():
    from nhmc import plugin, sampler
    assert list(inspect.signature(plugin.Base_Algo.__init__).parameters) == ['self', 'model', 'H_funcs', 'sigma_0', 'cls_fn']
    assert list(inspect.signature(plugin.HMC.cal_x0).parameters) == ['self', 'xt', 't', 'at', 'at_next', 'y_0', 'noise', 'classes']
    assert list(inspect.signature(plugin.HMC.map_back).parameters) == ['self', 'x0_t', 'y_0', 'add_up', 'at_next', 'at']
    assert list(inspect.signature(sampler.hmc).parameters) == ['x', 'n', 'b', 'seq', 'seq_next', 'algo', 'opt', 'y_0', 'H_funcs', 'x_orig']
    assert list(inspect.signature(sampler.iterative_sampling).parameters)[:8] == ['xt', 'n', 'b', 'seq', 'seq_next', 'algo', 'opt', 'y_0']
    with pytest.raises(TypeError):
        plugin.Base_Algo(None, None, 0.1)                                            # abstract, as in the reference
    for name in ('H', 'Ht', 'H_pinv', 'is_linear'):
        from nhmc import operators
        assert hasattr(operators.H_functions, name)


def test_hmc_refuses_operators_without_a_fused_data_term():
    from nhmc import plugin, sampler
    import types
    algo = plugin.HMC(lambda x, t: x, None, 0.1)
    opt = types.SimpleNamespace(tau=1.0, epsilon=0.05, m=1.0, sigma_0=0.1)
    with pytest.raises(TypeError, match='data_term'):
        sampler.hmc_chains(torch.zeros(1, 3, 8, 8), torch.zeros(1000), [250], [-1], algo, opt, torch.zeros(1, 4), object())


# ---- score network architecture ---------------------------------------------------------------------
def test_unet_matches_reference_architecture(golden):
    from nhmc import unet
    g = golden('g6_unet_64.npz')
    cfg = dict(image_size=64, num_channels=32, num_res_blocks=1, channel_mult='', learn_sigma=True,
               attention_resolutions='16', num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True)
    net = unet.create_model(**cfg).eval()
    keys = '\n'.join(f'{k} {tuple(v.shape)}' for k, v in net.state_dict().items())
    assert hashlib.sha256(keys.encode()).hexdigest() == str(g['keys_small_sha256'])
    gen = torch.Generator().manual_seed(int(g['weight_seed']))
    net.load_state_dict({k: torch.randn(v.shape, generator=gen) * float(g['weight_scale']) for k, v in net.state_dict().items()})
    with torch.no_grad():
        out = net(T(g['x']), T(g['t']))
    assert out.shape == (2, 6, 64, 64)
    assert float((out - T(g['out'])).abs().max()) <= 1e-5 * float(T(g['out']).abs().max())


def test_unet_ffhq_checkpoint_layout():
    from nhmc import unet
    with torch.device('meta'):
        net = unet.create_model(**unet.FFHQ_CONFIG)
    keys = '\n'.join(f'{k} {tuple(v.shape)}' for k, v in net.state_dict().items())
    from tests.conftest import load_golden
    g = load_golden('g6_unet_64.npz')
    assert hashlib.sha256(keys.encode()).hexdigest() == str(g['keys_ffhq_sha256'])
    assert sum(v.numel() for v in net.state_dict().values()) == int(g['n_params_ffhq'])
    with pytest.raises(NotImplementedError):
        unet.create_model(**{**unet.FFHQ_CONFIG, 'use_fp16': True})


# ---- oracle building blocks -------------------------------------------------------------------------
def test_philox_known_answers():
    """Random123 known-answer vectors for Philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox_ref.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(np.asarray(v).reshape(-1)[0]) for v in got) == want


def test_philox_normals_have_unit_moments():
    z = np.concatenate([philox_ref.randn_chain(11, c, 0, 3 * 64 * 64) for c in range(8)]).astype(np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.02 and abs((z ** 4).mean() - 3) < 0.1
    u = np.array([philox_ref.uniform_chain(11, c, d) for c in range(50) for d in range(20)])
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.03


def test_oracle_operators_are_adjoint_pairs():
    g_ = torch.Generator().manual_seed(3)
    dim = 32
    ops = [oops.InpaintRef(3, dim, oops.random_inpaint_missing(dim, generator=g_)), oops.BlockMeanRef(3, dim, 4),
           oops.SpectralBlurRef.from_kernels(oops.gaussian_taps(1.0), oops.gaussian_taps(20.0), 3, dim)]
    for op in ops:
        x = torch.randn(2, 3, dim, dim, generator=g_, dtype=torch.float32)
        y = torch.randn(2, op.M, generator=g_)
        lhs = (op.H(x).double() * y.double()).sum()
        rhs = (x.reshape(2, -1).double() * op.Ht(y).double()).sum()
        assert abs(float(lhs - rhs)) < 1e-3 * (1 + abs(float(lhs)))
