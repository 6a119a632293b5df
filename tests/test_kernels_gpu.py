"""GPU: every HIP kernel (through the C ABI via nhmc.kernels) against the CPU oracle.

Elementwise kernels are compiled with -ffp-contract=off and follow the reference's fp32 op
order, so they are required to be BIT-EXACT against the oracle; reductions (fp64 partials)
are required to agree to 1e-6 relative; the MFMA chain to 2e-5 relative (north_star: 1e-4).
"""
import numpy as np
import pytest
import torch

from oracle import ddim, hmc_ref, operators, philox_ref, schedule

pytestmark = pytest.mark.gpu

K = None


@pytest.fixture(scope='module', autouse=True)
def _kernels():
    global K
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    import nhmc.kernels as kernels
    K = kernels
    yield


def dev(t):
    return t.cuda().contiguous()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def gen(seed):
    return torch.Generator().manual_seed(seed)


def same_bits(a, b, name=''):
    """Bit-exact comparison with a useful message (count, worst ulp distance, first offender)."""
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if torch.equal(a, b):
        return True
    ia, ib = a.view(torch.int32).long().reshape(-1), b.view(torch.int32).long().reshape(-1)
    bad = torch.nonzero((a != b).reshape(-1)).reshape(-1)
    k = int(bad[0])
    raise AssertionError(f'{name}: {bad.numel()} of {a.numel()} elements differ; worst ulp distance '
                         f'{int((ia - ib).abs()[bad].max())}; first at {k}: got {a.reshape(-1)[k]!r} want {b.reshape(-1)[k]!r}')


SHAPES = [(1, 3, 16, 16), (3, 3, 32, 32), (2, 3, 256, 256)]


# ---- a1-a4: fused leapfrog ------------------------------------------------------------------
@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('mode', ['first', 'mid', 'last'])
def test_leapfrog_bit_exact(shape, mode):
    g_ = gen(1)
    x, p, g = (torch.randn(shape, generator=g_) for _ in range(3))
    B = shape[0]
    eps = np.array([0.05, 0.05 * 0.95, 0.01][:B])
    sig = np.array([1.7, 0.1, 0.6314][:B])
    m = 1.3
    rx, rp, rSx, rSp = hmc_ref.leapfrog_update(mode, x, p, g, eps=eps, sigma_y=sig, m=m)
    dx, dp, dg = dev(x), dev(p), dev(g)
    ws = K.leapfrog_ws(B, x[0].numel(), 'cuda')
    K.leapfrog_fused({'first': 0, 'mid': 1, 'last': 2}[mode], dx, dp, dg,
                     torch.tensor(eps, dtype=torch.float64, device='cuda'),
                     torch.tensor(sig, dtype=torch.float64, device='cuda'), m ** (-1), ws)
    assert torch.equal(dx.cpu(), rx) and torch.equal(dp.cpu(), rp)
    if mode != 'mid':
        tiles = K.leapfrog_tiles(x[0].numel())
        Sx = K.sum_partials(ws, tiles, B, stride=2, offset=0)
        Sp = K.sum_partials(ws, tiles, B, stride=2, offset=1)
        assert rel(Sx, rSx) < 1e-6 and rel(Sp, rSp) < 1e-6


def test_leapfrog_second_gradient_is_added_first():
    g_ = gen(2)
    x, p, ga, gb = (torch.randn(2, 3, 32, 32, generator=g_) for _ in range(4))
    rx, rp, _, _ = hmc_ref.leapfrog_update('mid', x, p, ga + gb, eps=0.05, sigma_y=0.9, m=1.0)
    dx, dp = dev(x), dev(p)
    K.leapfrog_fused(1, dx, dp, dev(ga), 0.05, 0.9, 1.0, g2=dev(gb))
    assert torch.equal(dx.cpu(), rx) and torch.equal(dp.cpu(), rp)


def test_leapfrog_rejects_bad_operands():
    import nhmc._lib as L
    x = torch.zeros(1, 3, 5, 5, device='cuda')          # 75 elements: not a multiple of 4
    with pytest.raises(L.NhmcError):
        K.leapfrog_fused(1, x, x.clone(), x.clone(), 0.1, 1.0, 1.0)
    with pytest.raises(L.NhmcError):
        K.leapfrog_fused(1, torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4), 0.1, 1.0, 1.0)


# ---- a8-a11: DDIM mix -----------------------------------------------------------------------
def _alphas(B):
    b = schedule.betas_fp32()
    t = torch.tensor([750, 500, 250][:B] if B <= 3 else [750] * B)
    tn = torch.tensor([500, 250, -1][:B] if B <= 3 else [500] * B)
    return schedule.alpha_bar(b, t), schedule.alpha_bar(b, tn)


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('ech', [3, 6])
def test_ddim_mix_forward_bit_exact(shape, ech):
    g_ = gen(3)
    B = shape[0]
    xt = torch.randn(shape, generator=g_)
    e = torch.randn((B, ech) + shape[2:], generator=g_)
    at, atn = _alphas(B)
    x0, add = ddim.predict_x0(xt, e, at, atn)
    nxt = ddim.renoise(x0, add, atn)
    out = K.ddim_mix_fwd(dev(xt), dev(e), at, atn, want=('xt_next', 'x0_t', 'add_up'))
    same_bits(out['x0_t'], x0, 'x0_t')
    same_bits(out['add_up'], add, 'add_up')
    same_bits(out['xt_next'], nxt, 'xt_next')
    only = K.ddim_mix_fwd(dev(xt), dev(e), at, atn, final_clip=True)
    same_bits(only['xt_next'], nxt.clip(-1, 1), 'clipped')
    same_bits(K.ddim_map_back(dev(x0), dev(add), atn), nxt, 'map_back')


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('final_clip', [False, True])
def test_ddim_mix_backward_matches_autograd_bit_exact(shape, final_clip):
    g_ = gen(4)
    B = shape[0]
    xt = torch.randn(shape, generator=g_).requires_grad_(True)
    e = torch.randn((B, 6) + shape[2:], generator=g_).requires_grad_(True)
    gout = torch.randn(shape, generator=g_)
    at, atn = _alphas(B)
    out = ddim.ddim_step(xt, e, at, atn)
    if final_clip:
        out = out.clip(-1, 1)
    ga, gb = torch.autograd.grad(out, (xt, e), gout)
    dx, de = K.ddim_mix_bwd(dev(gout), dev(xt.detach()), dev(e.detach()), at, atn, final_clip=final_clip)
    same_bits(dx, ga, 'g_xt')
    same_bits(de, gb, 'g_e')
    assert float(de[:, 3:].abs().max()) == 0.0


def test_ddim_mix_backward_sums_two_upstream_gradients():
    g_ = gen(5)
    xt, g1, g2 = (torch.randn(2, 3, 32, 32, generator=g_) for _ in range(3))
    e = torch.randn(2, 6, 32, 32, generator=g_)
    at, atn = _alphas(2)
    a = K.ddim_mix_bwd(dev(g1 + g2), dev(xt), dev(e), at, atn)
    b = K.ddim_mix_bwd(dev(g1), dev(xt), dev(e), at, atn, gout2=dev(g2))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


# ---- a12-a14: data terms and operator surface -----------------------------------------------
def _inpaint(dim, seed=0):
    missing = operators.random_inpaint_missing(dim, generator=gen(seed))
    ref = operators.InpaintRef(3, dim, missing)
    import nhmc.operators as ops
    return ref, ops.Inpainting(3, dim, missing, 'cuda')


@pytest.mark.parametrize('dim,B', [(16, 1), (32, 3), (256, 2)])
def test_inpaint_data_term(dim, B):
    ref, op = _inpaint(dim)
    g_ = gen(6)
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8          # some |xt| > 1: clip mask matters
    y = torch.randn(B, ref.M, generator=g_)
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    loss, g = K.data_inpaint(dev(xt), dev(y), op.slot, apply_clip=True)
    same_bits(g, g_ref, 'g_xt')
    assert rel(loss, loss_ref) < 1e-6
    assert torch.equal(op.H(dev(xt)).cpu(), ref.H(xt))
    assert torch.equal(op.Ht(dev(y)).cpu(), ref.Ht(y))
    assert torch.equal(op.H_pinv(dev(y)).cpu(), ref.H_pinv(y))


@pytest.mark.parametrize('dim,r,B', [(32, 4, 3), (32, 2, 1), (64, 16, 2), (256, 4, 2), (256, 16, 1), (64, 8, 1), (64, 32, 1)])
def test_sr_data_term(dim, r, B):
    import nhmc.operators as ops
    ref, op = operators.BlockMeanRef(3, dim, r), ops.SuperResolution(3, dim, r, 'cuda')
    g_ = gen(7)
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(B, ref.M, generator=g_)
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    loss, g = K.data_sr(dev(xt), dev(y), r, apply_clip=True)
    assert rel(g, g_ref) < 2e-6 and rel(loss, loss_ref) < 2e-6
    assert rel(op.H(dev(xt)), ref.H(xt)) < 2e-6
    assert torch.equal(op.Ht(dev(y)).cpu(), ref.Ht(y))
    assert torch.equal(op.H_pinv(dev(y)).cpu(), ref.H_pinv(y))


def _aniso(dim):
    import nhmc.operators as ops
    ref = operators.SpectralBlurRef.from_kernels(operators.gaussian_taps(1.0), operators.gaussian_taps(20.0), 3, dim)
    return ref, ops.Deblurring2D.from_factors(ref.U1, ref.U2, ref.V1, ref.V2, ref.D, 'cuda')


@pytest.mark.parametrize('dim,B', [(32, 2), (64, 3), (256, 2)])
def test_spectral_operator_and_data_term(dim, B):
    ref, op = _aniso(dim)
    g_ = gen(8)
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(B, ref.M, generator=g_)
    assert rel(op.H(dev(xt)), ref.H(xt)) < 2e-5
    assert rel(op.Ht(dev(y)), ref.Ht(y)) < 2e-5
    assert rel(op.H_pinv(dev(y)), ref.H_pinv(y)) < 2e-5
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    yT = dev(y).reshape(B, 3, dim, dim).transpose(-1, -2).contiguous()          # the kernel takes the planes transposed
    loss, g = K.data_spectral(dev(xt), yT, op.factors, op.Dmap, apply_clip=True, DmapT=op.DmapT)
    assert rel(loss, loss_ref) < 2e-5 and rel(g, g_ref) < 2e-5
    loss_op, g_op = op.data_term(dev(xt), dev(y), apply_clip=True)             # the operator does that transposition itself
    assert torch.equal(g_op, g) and torch.equal(loss_op, loss)


@pytest.mark.parametrize('dim', [32, 64])
def test_spectral_data_term_is_the_reference_arithmetic_bit_for_bit(dim):
    """Every MFMA product is an exact k-ascending FMA chain from zero -- what torch's CPU sgemm computes (tools/mfma_bits.py,
    K <= 64 on the GPU box's host; up to 256 on the build container's) -- the forward multiplies by the left factor first
    as Hfuncs.py:493-509 does and the adjoint by the right factor first as autograd does: the gradient of the data term is
    the oracle's autograd gradient, bit for bit (the loss differs in its summation only)."""
    ref, op = _aniso(dim)
    g_ = gen(80 + dim)
    B = 3
    xt = torch.randn(B, 3, dim, dim, generator=g_) * 0.8
    y = torch.randn(B, ref.M, generator=g_)
    assert torch.equal(op.H(dev(xt)).cpu(), ref.H(xt)) and torch.equal(op.Ht(dev(y)).cpu(), ref.Ht(y))
    loss_ref, g_ref = hmc_ref.data_term(xt, ref, y)
    loss, g = op.data_term(dev(xt), dev(y), apply_clip=True)
    assert torch.equal(g.cpu(), g_ref), float((g.cpu() - g_ref).abs().max() / g_ref.abs().max())
    assert rel(loss, loss_ref) < 1e-6


def test_spectral_against_reference_operator_data(golden):
    """Operator data exported from the reference object (64x64) + reference outputs."""
    import nhmc.operators as ops
    g = golden('g2_ops_64.npz')
    T = torch.from_numpy
    op = ops.Deblurring2D.from_factors(*(T(g[f'aniso_{k}']) for k in ('U1', 'U2', 'V1', 'V2', 'D')), 'cuda')
    assert rel(op.H(dev(T(g['x']))), T(g['aniso_Hx'])) < 2e-5
    assert rel(op.Ht(dev(T(g['aniso_y']))), T(g['aniso_Hty'])) < 2e-5
    assert rel(op.H_pinv(dev(T(g['aniso_y']))), T(g['aniso_Hpinvy'])) < 5e-5


@pytest.mark.parametrize('name', ['inpaint', 'sr4', 'sr16'])
def test_operators_against_reference_goldens(golden, name):
    import nhmc.operators as ops
    g = golden('g2_ops_64.npz')
    T = torch.from_numpy
    op = ops.Inpainting(3, 64, T(g['missing']), 'cuda') if name == 'inpaint' else \
        ops.SuperResolution(3, 64, int(name[2:]), 'cuda')
    tol = 0.0 if name == 'inpaint' else 2e-6
    assert rel(op.H(dev(T(g['x']))), T(g[f'{name}_Hx'])) <= tol
    assert rel(op.Ht(dev(T(g[f'{name}_y']))), T(g[f'{name}_Hty'])) <= tol
    assert rel(op.H_pinv(dev(T(g[f'{name}_y']))), T(g[f'{name}_Hpinvy'])) <= tol


# ---- a5-a7: Hamiltonian, Metropolis, schedules ------------------------------------------------
def test_hamiltonian_matches_reference_fp32_order():
    g_ = gen(9)
    B, shape = 3, (3, 3, 64, 64)
    x, p = torch.randn(shape, generator=g_), torch.randn(shape, generator=g_)
    loss = torch.rand(B, generator=g_).double() * 5e3
    sig = np.array([1.7, 0.1, 0.35])
    m = 1.0
    ws = K.leapfrog_ws(B, x[0].numel(), 'cuda')
    dx, dp = dev(x), dev(p)
    K.leapfrog_fused(0, dx, dp, dev(torch.zeros(shape)), 0.0, sig.tolist(), 1.0, ws)       # eps = 0: sums only
    H, terms = K.hamiltonian(ws, x[0].numel(), dev(loss), sig.tolist(), m ** (-1), want_terms=True)
    Sx = torch.sum(x ** 2, dim=(1, 2, 3))
    Sp = torch.sum(p * p, dim=(1, 2, 3))
    Href = hmc_ref.hamiltonian(Sx, loss.float(), Sp, 1 / (2 * sig ** 2), m)
    assert rel(terms[:, 0], Sx) < 1e-6 and rel(terms[:, 1], Sp) < 1e-6
    # one fp32 ulp of H (~1e4) is ~1e-3; allow two
    assert float((H.cpu() - Href).abs().max()) <= 2 * float(np.spacing(np.float32(Href.abs().max())))


def test_metropolis_and_schedule_follow_the_reference_rules():
    H0 = torch.tensor([10.0, 10.0, 10.0, 10.0, float('nan')])
    H1 = torch.tensor([9.0, 10.5, 12.0, 10.5, 1.0])
    u = torch.tensor([0.99, 0.5, 0.5, 0.7, 0.0])
    active = torch.tensor([1, 1, 1, 0, 1], dtype=torch.int32)
    acc, dH = K.metropolis(dev(H0), dev(H1), dev(u), dev(active))
    # exp(-0.5)=0.6065 > 0.5 accept; exp(-2)=0.135 < 0.5 reject; inactive never accepts; NaN rejects
    assert acc.cpu().tolist() == [1, 1, 0, 0, 0]
    assert torch.allclose(dH.cpu()[:4], (H1 - H0)[:4])

    B, epochs, sampling, sigma_0 = 6, 60, 20, 0.1
    st = dict(epoch=torch.tensor([0, 30, 59, 60, 61, 100], dtype=torch.int32),
              rejected=torch.tensor([0, 1, 0, 2, 0, 0], dtype=torch.int32),
              tau=torch.full((B,), 1.0, dtype=torch.float64), eps=torch.full((B,), 0.05, dtype=torch.float64),
              sigma_y=torch.full((B,), 0.123, dtype=torch.float64), eps_eff=torch.zeros(B, dtype=torch.float64),
              active=torch.zeros(B, dtype=torch.int32), n_accept=torch.zeros(B, dtype=torch.int32),
              n_reject=torch.zeros(B, dtype=torch.int32))
    st = {k: v.cuda() for k, v in st.items()}
    K.schedule_begin(st, sigma_0, epochs, sampling)
    sy = st['sigma_y'].cpu().tolist()
    assert sy[0] == hmc_ref.sigma_y_at(0, sigma_0) and sy[1] == hmc_ref.sigma_y_at(30, sigma_0)
    assert sy[2] == hmc_ref.sigma_y_at(59, sigma_0) and sy[3] == sigma_0 and sy[4] == 0.123
    assert st['tau'].cpu().tolist() == [1.0, 1.0, 1.0, 0.1, 1.0, 1.0]
    assert st['eps'].cpu().tolist() == [0.05, 0.05, 0.05, 0.01, 0.05, 0.05]
    assert st['active'].cpu().tolist() == [1, 1, 1, 1, 1, 0]
    assert st['eps_eff'].cpu().tolist() == [0.05, 0.05, 0.05, 0.01, 0.05, 0.0]
    accept = torch.tensor([1, 0, 0, 0, 1, 1], dtype=torch.int32).cuda()
    K.schedule_end(accept, st)
    assert st['epoch'].cpu().tolist() == [1, 30, 59, 60, 62, 100]
    assert st['rejected'].cpu().tolist() == [0, 2, 1, 3, 0, 0]
    assert st['tau'].cpu().tolist() == [1.0, 1.0 * 0.95, 1.0, 0.1 * 0.95, 1.0, 1.0]
    assert st['eps'].cpu().tolist() == [0.05, 0.05 * 0.95, 0.05, 0.01 * 0.95, 0.05, 0.05]


def test_accept_commit_moves_only_accepted_chains_and_collects_samples():
    B, shape, epochs, sampling = 3, (3, 3, 16, 16), 4, 2
    g_ = gen(10)
    x, xp, xtp = (torch.randn(shape, generator=g_) for _ in range(3))
    samples = torch.zeros(B, sampling, *shape[1:])
    accept = torch.tensor([1, 0, 1], dtype=torch.int32)
    epoch = torch.tensor([3, 7, 7], dtype=torch.int32)          # slots: -3 (none), -, 1
    dx, ds = dev(x), dev(samples)
    K.accept_commit(dev(accept), dev(epoch), dx, dev(xp), dev(xtp), ds, epochs, sampling)
    assert torch.equal(dx.cpu()[0], xp[0]) and torch.equal(dx.cpu()[1], x[1]) and torch.equal(dx.cpu()[2], xp[2])
    assert float(ds[0].abs().max()) == 0 and float(ds[1].abs().max()) == 0
    assert torch.equal(ds.cpu()[2, 1], xtp[2]) and float(ds[2, 0].abs().max()) == 0


def test_psnr_matches_reference_formula():
    g_ = gen(11)
    a, b = torch.randn(2, 3, 64, 64, generator=g_), torch.rand(2, 3, 64, 64, generator=g_) * 2 - 1
    ref = torch.stack([hmc_ref.psnr_unit(hmc_ref.to_unit_range(a[i]), hmc_ref.to_unit_range(b[i])) for i in range(2)])
    assert torch.allclose(K.psnr(dev(a), dev(b)).cpu(), ref, rtol=0, atol=1e-4)


# ---- a1: counter-based noise ----------------------------------------------------------------
def test_philox_matches_numpy_restatement_and_is_shard_invariant():
    seed, n = 5678 | (77 << 32), 3 * 32 * 32
    out = K.randn_philox((4, 3, 32, 32), seed, chain_id0=10, draw=3).cpu().numpy().reshape(4, -1)
    for c in range(4):
        ref = philox_ref.randn_chain(seed, 10 + c, 3, n)
        assert np.abs(out[c] - ref).max() < 2e-5
    # the same chains generated as two shards
    lo = K.randn_philox((2, 3, 32, 32), seed, 10, 3).cpu().numpy().reshape(2, -1)
    hi = K.randn_philox((2, 3, 32, 32), seed, 12, 3).cpu().numpy().reshape(2, -1)
    assert np.array_equal(np.concatenate([lo, hi]), out)
    u = K.uniform_philox(4, seed, 10, 3).cpu().numpy()
    assert np.allclose(u, [philox_ref.uniform_chain(seed, 10 + c, 3) for c in range(4)], rtol=0, atol=1e-7)
    big = K.randn_philox((8, 3, 256, 256), 1, 0, 0).double()
    assert abs(float(big.mean())) < 3e-3 and abs(float(big.var()) - 1) < 5e-3
    assert abs(float((big ** 4).mean()) - 3) < 5e-2


# ---- full-size properties (BASELINE config 1: B=64, 3x256x256) -------------------------------
def test_full_size_properties_b64():
    B, shape = 64, (64, 3, 256, 256)
    x = K.randn_philox(shape, 5678, 0, 0)
    p = K.randn_philox(shape, 5678, 0, 1)
    g = K.randn_philox(shape, 5678, 0, 2)
    # (1) eps = 0 is the identity on (x, p)
    x0, p0 = x.clone(), p.clone()
    K.leapfrog_fused(1, x, p, g, 0.0, 1.0, 1.0)
    assert torch.equal(x, x0) and torch.equal(p, p0)
    # (2) a few chains of the full batch against the oracle, bit for bit
    K.leapfrog_fused(1, x, p, g, 0.05, 0.1, 1.0)
    for c in (0, 31, 63):
        rx, rp, _, _ = hmc_ref.leapfrog_update('mid', x0[c:c + 1].cpu(), p0[c:c + 1].cpu(), g[c:c + 1].cpu(),
                                               eps=0.05, sigma_y=0.1, m=1.0)
        assert torch.equal(x[c:c + 1].cpu(), rx) and torch.equal(p[c:c + 1].cpu(), rp)
    # (3) operator adjointness <Hx, y> = <x, H^T y> at full size, and H H^+ y = y for inpainting
    ref, op = _inpaint(256, seed=5678)
    y = torch.randn(B, ref.M, device='cuda', generator=torch.Generator(device='cuda').manual_seed(9))
    lhs = (op.H(x).double() * y.double()).sum()
    rhs = (x.reshape(B, -1).double() * op.Ht(y).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-9 * abs(float(lhs)) + 1e-6
    assert torch.equal(op.H(op.H_pinv(y).reshape(shape)), y)
    # (4) data-term gradient is -2 H^T (y - H clip x) masked, loss is its norm
    loss, gx = K.data_inpaint(x, y, op.slot, apply_clip=True)
    r = y - op.H(x.clip(-1, 1))
    assert rel(loss, (r.double() ** 2).sum(1)) < 1e-7
    mask = ((x >= -1) & (x <= 1)).float().reshape(B, -1)
    assert torch.equal(gx.reshape(B, -1), -(2 * op.Ht(r)) * mask)
