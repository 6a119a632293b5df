"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python oracle/gen_golden.py            # needs /root/reference; never runs on the GPU box

The reference ships no fixtures for the HMC path, so the
oracle is pinned against outputs of the reference itself.  This script imports
the reference's own modules from /root/reference (read-only; bytecode writing
disabled), feeds them seeded inputs and stores inputs + outputs as plain arrays.
Packages the reference imports at module level but never touches on this path
(torchvision, skimage, lpips) are absent from the image; inert placeholder
modules satisfy those imports (recipe: SURVEY.md Appendix A).  Only data is
written to the repo; no reference source is copied.

Fixtures
  g1_schedule.npz          alpha-bar table + the four constants used by timesteps=3
  g2_ops_{32,64}.npz       H / Ht / H_pinv of Inpainting, SuperResolution(4[,16]), Deblurring2D
                           (+ the aniso operator data U1,U2,V1,V2,D exported from the reference object)
  g3_ddim_32.npz           iterative_sampling output + autograd gradient for a seeded input
  g4_hmc_{inpaint,sr4,aniso}_32.npz
                           full `hmc()` run at 32x32 with the tiny score net: inputs, returned
                           [20,3,32,32], per-iteration -dH and u, printed PSNRs, and the first
                           trajectory's momentum, positions and decode
  g5_ops_256.npz           sparse probes of the three operators at 256x256
"""
import argparse
import contextlib
import io
import os
import re
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    def placeholder(name, **attrs):
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        sys.modules[name] = mod
        return mod

    tv = placeholder('torchvision')
    tv.utils = placeholder('torchvision.utils', save_image=lambda *a, **k: None, make_grid=None)
    tv.transforms = placeholder('torchvision.transforms')
    tv.transforms.functional = placeholder('torchvision.transforms.functional')
    tv.datasets = placeholder('torchvision.datasets')
    tv.datasets.utils = placeholder('torchvision.datasets.utils', verify_str_arg=None, iterable_to_str=None)
    placeholder('skimage').metrics = placeholder('skimage.metrics', structural_similarity=None)
    placeholder('lpips', LPIPS=None)
    import main_sampling as ms
    ms.device = torch.device('cpu')
    ms.config = ms.dict2namespace({'data': {'rescaled': True, 'logit_transform': False}})
    return ms


def np32(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(torch_version=np.array(torch.__version__), generator=np.array('oracle/gen_golden.py'))
    np.savez_compressed(os.path.join(OUT, name), **arrays, **meta)
    size = os.path.getsize(os.path.join(OUT, name))
    print(f'wrote {name}  ({size / 1024:.1f} KiB)')


# --------------------------------------------------------------------------- #
def g1_schedule(ms):
    betas = ms.get_beta_schedule(beta_schedule='linear', beta_start=1e-4, beta_end=0.02,
                                 num_diffusion_timesteps=1000)
    b = torch.from_numpy(betas).float()
    t = torch.arange(-1, 1000)
    table = ms.compute_alpha(b, t).reshape(-1)
    four = ms.compute_alpha(b, torch.tensor([750, 500, 250, -1])).reshape(-1)
    save('g1_schedule.npz', betas64=betas, table=np32(table), at_750_500_250_m1=np32(four))


def aniso_kernels():
    def taps(sigma):
        pdf = lambda x: torch.exp(torch.Tensor([-0.5 * (x / sigma) ** 2]))
        k = torch.Tensor([pdf(i) for i in range(-4, 5)])
        return k / k.sum()
    return taps(1), taps(20)          # kernel1 (sigma=1), kernel2 (sigma=20): main_sampling.py:327-335


def build_ops(ms, dim, seed):
    from obs_functions.Hfuncs import Inpainting, SuperResolution, Deblurring2D
    g = torch.Generator().manual_seed(seed)
    r = 3 * torch.randperm(dim * dim, generator=g)[: int(dim * dim * 0.92)].long()
    missing = torch.cat([r, r + 1, r + 2], dim=0)
    k1, k2 = aniso_kernels()
    ops = dict(inpaint=Inpainting(3, dim, missing, 'cpu'),
               sr4=SuperResolution(3, dim, 4, 'cpu'),
               aniso=Deblurring2D(k1, k2, 3, dim, 'cpu'))
    if dim % 16 == 0:
        ops['sr16'] = SuperResolution(3, dim, 16, 'cpu')
    return ops, missing


def export_aniso(db):
    hw = db.img_dim ** 2
    sing = db.singulars()
    D = torch.zeros(3, hw)
    for c in range(3):
        D[c, db._perm] = sing[3 * torch.arange(hw) + c]
    return dict(U1=np32(db.U_small1), U2=np32(db.U_small2), V1=np32(db.V_small1), V2=np32(db.V_small2),
                D=np32(D.reshape(3, db.img_dim, db.img_dim)), perm=np32(db._perm),
                s_sorted=np32(db._singulars))


def g2_ops(ms, dim):
    ops, missing = build_ops(ms, dim, seed=100 + dim)
    g = torch.Generator().manual_seed(7 + dim)
    x = torch.randn(2, 3, dim, dim, generator=g)
    arrays = dict(x=np32(x), missing=np32(missing))
    for name, op in ops.items():
        hx = op.H(x)
        y = torch.randn(hx.shape, generator=g)
        arrays[f'{name}_Hx'] = np32(hx)
        arrays[f'{name}_y'] = np32(y)
        arrays[f'{name}_Hty'] = np32(op.Ht(y.clone()))
        arrays[f'{name}_Hpinvy'] = np32(op.H_pinv(y.clone()))
    for k, v in export_aniso(ops['aniso']).items():
        arrays[f'aniso_{k}'] = v
    save(f'g2_ops_{dim}.npz', **arrays)


def tiny_model():
    from oracle.tiny_score import make_tiny_score
    net = make_tiny_score()
    path = os.path.join(OUT, 'tiny_score.pt')
    if not os.path.exists(path):
        os.makedirs(OUT, exist_ok=True)
        torch.save(net.state_dict(), path)
    return net


def g3_ddim(ms, dim=32):
    from algos.unconditional import Unconditional
    net = tiny_model()
    b = torch.from_numpy(ms.get_beta_schedule(beta_schedule='linear', beta_start=1e-4, beta_end=0.02,
                                              num_diffusion_timesteps=1000)).float()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, dim, dim, generator=g).requires_grad_(True)
    w = torch.randn(2, 3, dim, dim, generator=g)
    opt = types.SimpleNamespace(algo='hmc', noise='ddpm')
    algo = Unconditional(net, None, 0.1)
    xt = ms.iterative_sampling(x, 2, b, [250, 500, 750], [-1, 250, 500], algo, opt, None, tqdm_disable=True)
    grad = torch.autograd.grad((xt.clip(-1, 1) * w).sum(), x)[0]
    save(f'g3_ddim_{dim}.npz', x=np32(x), w=np32(w), xt=np32(xt), grad=np32(grad))


def g4_hmc(ms, deg, dim=32, seed=5678, f64=False, op=None, out_name=None, extra=None, probe=0, grid_bits=0, grid_absolute=False):
    """f64=True -> g14_hmc_f64_*: the same reference run with the tiny score evaluated in float64 (oracle.tiny_score.F64Score,
    the model is hmc()'s argument); stores every uniform and -dH so a GPU test can replay the whole run on the same tape.
    op / out_name / extra: another reference operator object, the fixture's file name and its operator data
    (oracle/gen_golden_hmc_ops2.py -> g15_*).  probe > 0 (full-size runs, oracle/gen_golden_hmc_256.py -> g16_*): the seeded
    inputs are NOT stored (the test regenerates x_orig, the observation noise and x from torch.Generator(11) in this
    function's order, and the mask from its seed) and the returned images are stored as `probe` positions + norms."""
    from algos.unconditional import Unconditional
    if op is None:
        ops, missing = build_ops(ms, dim, seed=900 + dim)
        Hf = ops[deg]
    else:
        Hf, missing = op, torch.zeros(0, dtype=torch.long)
    net = tiny_model()
    if f64 and grid_bits:                                     # oracle.tiny_score.GridF64Score: reproducible across devices (g16b)
        from oracle.tiny_score import GridF64Score
        net = GridF64Score(net, grid_bits, absolute=grid_absolute)
    elif f64:
        from oracle.tiny_score import F64Score
        net = F64Score(net)
    b = torch.from_numpy(ms.get_beta_schedule(beta_schedule='linear', beta_start=1e-4, beta_end=0.02,
                                              num_diffusion_timesteps=1000)).float()
    g = torch.Generator().manual_seed(11)
    x_orig = torch.rand(1, 3, dim, dim, generator=g) * 2 - 1
    sigma_0 = 2 * 0.05                                        # main_sampling.py:348
    y_0 = Hf.H(x_orig).detach()
    y_0 = y_0 + sigma_0 * torch.randn(y_0.shape, generator=g)
    x = torch.randn(1, 3, dim, dim, generator=g)
    opt = argparse.Namespace(tau=1.0, epsilon=0.05, m=1.0, sigma_0=sigma_0, algo='hmc', noise='ddpm',
                             image_folder='/tmp/nhmc_golden_scratch')
    os.makedirs(opt.image_folder, exist_ok=True)
    algo = Unconditional(net, Hf, sigma_0)

    # --- recorders (wrap, never alter) ------------------------------------- #
    rec = dict(p=[], u=[], neg_dH=[], pos=[], dec=[])
    real_randn_like, real_rand, real_exp, real_iter = torch.randn_like, torch.rand, torch.exp, ms.iterative_sampling

    def randn_like(*a, **k):
        out = real_randn_like(*a, **k)
        if len(rec['p']) < 1:
            rec['p'].append(out.clone())
        rec['p_last'] = [out.clone()]
        return out

    def rand(*a, **k):
        out = real_rand(*a, **k)
        rec['u'].append(float(out.reshape(-1)[0]))
        return out

    def exp(t, *a, **k):
        if t.numel() == 1 and t.dim() == 1:
            rec['neg_dH'].append(float(t.reshape(-1)[0]))
        return real_exp(t, *a, **k)

    def iter_sampling(xin, *a, **k):
        out = real_iter(xin, *a, **k)
        if len(rec['pos']) < 21:
            rec['pos'].append(xin.detach().clone())
            rec['dec'].append(out.detach().clone())
        return out

    torch.manual_seed(seed)
    buf = io.StringIO()
    torch.randn_like, torch.rand, torch.exp, ms.iterative_sampling = randn_like, rand, exp, iter_sampling
    try:
        with contextlib.redirect_stdout(buf):
            out = ms.hmc(x, 1, b, [250, 500, 750], [-1, 250, 500], algo, opt, y_0, Hf, x_orig)
    finally:
        torch.randn_like, torch.rand, torch.exp, ms.iterative_sampling = real_randn_like, real_rand, real_exp, real_iter
    psnr = [float(v) for v in re.findall(r'PSNR: ([-0-9.e+inf]+)', buf.getvalue())]
    arrays = dict(x=np32(x), y_0=np32(y_0), x_orig=np32(x_orig), missing=np32(missing),
                  seed=np.array(seed), sigma_0=np.array(sigma_0), tau=np.array(1.0), epsilon=np.array(0.05),
                  m=np.array(1.0), out=np32(out), psnr=np.array(psnr), u=np.array(rec['u']),
                  neg_dH=np.array(rec['neg_dH']), p0=np32(rec['p'][0]),
                  pos_first=np32(rec['pos'][1]), pos_last=np32(rec['pos'][20]), dec_last=np32(rec['dec'][20]),
                  dec_init=np32(rec['dec'][0]))
    if deg == 'aniso':
        for k, v in export_aniso(Hf).items():
            arrays[f'aniso_{k}'] = v
    if f64:
        for k in ('pos_first', 'pos_last', 'dec_last', 'dec_init'):
            arrays.pop(k)
        arrays['p_last'] = np32(rec['p_last'][0])
    if probe:
        flat = out.reshape(out.shape[0], -1)
        pos = torch.randperm(flat.shape[1], generator=torch.Generator().manual_seed(16))[:probe]
        for k in ['x', 'x_orig', 'missing', 'out', 'p0', 'p_last'] + [k for k in arrays if k.startswith('aniso_')]:
            arrays.pop(k, None)                                    # the aniso instance is G13's (checked by the caller)
        arrays.update(out_probe_pos=np32(pos).astype(np.int32), out_probe=np32(flat[:, pos]), out_norm=np32(flat.double().norm(dim=1)),
                      out_absmax=np32(flat.abs().max()), missing_sum=np.array(int(missing.sum())), mask_seed=np.array(900 + dim),
                      p0_head=np32(rec['p'][0].reshape(-1)[:64]), p_last_head=np32(rec['p_last'][0].reshape(-1)[:64]),
                      x_head=np32(x.reshape(-1)[:64]), x_orig_head=np32(x_orig.reshape(-1)[:64]))
    arrays.update(extra or {})
    if grid_bits:
        arrays['grid_bits'] = np.array(grid_bits)
        arrays['grid_absolute'] = np.array(int(grid_absolute))
    save(out_name or (f'g14_hmc_f64_{deg}_{dim}.npz' if f64 else f'g4_hmc_{deg}_{dim}.npz'), **arrays)
    print(f'   {deg}: {len(rec["u"])} iterations, {len(psnr)} accepts, final PSNR {psnr[-1]:.3f}')


def g5_ops_256(ms):
    ops, missing = build_ops(ms, 256, seed=5678)
    g = torch.Generator().manual_seed(256)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    arrays = dict(x_seed=np.array(256), missing=np32(missing).astype(np.int32))
    for name, op in ops.items():
        hx = op.H(x)
        y = torch.randn(hx.shape, generator=torch.Generator().manual_seed(99))
        hty = op.Ht(y.clone())
        pm = torch.randperm(hx.shape[1], generator=torch.Generator().manual_seed(1))[:256]
        pn = torch.randperm(hty.shape[1], generator=torch.Generator().manual_seed(2))[:256]
        arrays[f'{name}_probe_m'] = np32(pm)
        arrays[f'{name}_probe_n'] = np32(pn)
        arrays[f'{name}_Hx_probe'] = np32(hx[0, pm])
        arrays[f'{name}_Hx_norm'] = np32(hx.double().norm())
        arrays[f'{name}_Hty_probe'] = np32(hty[0, pn])
        arrays[f'{name}_Hty_norm'] = np32(hty.double().norm())
    save('g5_ops_256.npz', **arrays)


def g13_aniso_256(ms):
    """deblur_aniso at BASELINE size: the reference object's operator data (U1,U2,V1,V2 and the sort permutation, from
    which D follows) and probes of H, Ht, H_pinv -- what pins configs[3]'s operator position by position."""
    from obs_functions.Hfuncs import Deblurring2D
    k1, k2 = aniso_kernels()
    torch.set_num_threads(4)      # LAPACK's blocking -- hence the near-null singular vectors -- depends on it; G5 used 4 too
    db = Deblurring2D(k1, k2, 3, 256, 'cpu')
    ex = export_aniso(db)
    g = torch.Generator().manual_seed(1313)
    x = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    y = torch.randn(2, 3 * 256 * 256, generator=g)
    hx, hty, hpy = db.H(x), db.Ht(y.clone()), db.H_pinv(y.clone()).reshape(2, -1)
    pm = torch.randperm(hx.shape[1], generator=g)[:512]
    arrays = dict(U1=ex['U1'], U2=ex['U2'], V1=ex['V1'], V2=ex['V2'], perm=ex['perm'].astype(np.uint16),
                  s1=np32(db.singulars_small1), s2=np32(db.singulars_small2), s_sorted=ex['s_sorted'],
                  kernel1=np32(k1), kernel2=np32(k2), xy_seed=np.array(1313), num_threads=np.array(torch.get_num_threads()), probe=np32(pm).astype(np.int32),
                  Hx_probe=np32(hx[:, pm]), Hty_probe=np32(hty[:, pm]), Hpinv_probe=np32(hpy[:, pm]),
                  Hx_norm=np32(hx.double().norm(dim=1)), Hty_norm=np32(hty.double().norm(dim=1)),
                  Hpinv_norm=np32(hpy.double().norm(dim=1)))
    assert int(ex['perm'].max()) < 65536
    save('g13_aniso_256.npz', **arrays)


if __name__ == '__main__' and 'g14' in sys.argv[1:]:
    _ms = import_reference()
    torch.set_num_threads(4)
    for _deg in ('inpaint', 'sr4', 'aniso'):
        g4_hmc(_ms, _deg, f64=True)
    sys.exit(0)

if __name__ == '__main__' and 'g13' in sys.argv[1:]:
    g13_aniso_256(import_reference())
    sys.exit(0)

if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5']
    ms = import_reference()
    torch.set_num_threads(4)
    if 'g1' in which:
        g1_schedule(ms)
    if 'g2' in which:
        g2_ops(ms, 32)
        g2_ops(ms, 64)
    if 'g3' in which:
        g3_ddim(ms)
    if 'g4' in which:
        for deg in ('inpaint', 'sr4', 'aniso'):
            g4_hmc(ms, deg)
    if 'g5' in which:
        g5_ops_256(ms)


def g6_unet(ms):
    """Reference U-Net architecture (guided_diffusion/unet_ffhq.py) at a reduced width with seeded weights:
    pins nhmc.unet's module tree (state_dict keys, in order) and forward arithmetic."""
    from guided_diffusion.unet_ffhq import create_model
    import hashlib
    cfg = dict(image_size=64, num_channels=32, num_res_blocks=1, channel_mult='', learn_sigma=True, class_cond=False,
               use_checkpoint=False, attention_resolutions='16', num_heads=4, num_head_channels=16,
               num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0, resblock_updown=True, use_fp16=False,
               use_new_attention_order=False, model_path='')
    with contextlib.redirect_stdout(io.StringIO()):
        net = create_model(**cfg).eval()
    g = torch.Generator().manual_seed(606)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    x = torch.randn(2, 3, 64, 64, generator=g)
    t = torch.tensor([750.0, 250.0])
    with torch.no_grad():
        out = net(x, t)
    gout = torch.randn(out.shape, generator=g)                     # input gradient (what the HMC backward asks of the net)
    xl = x.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(net(xl, t), xl, gout)
    with contextlib.redirect_stdout(io.StringIO()):
        full = create_model(**{**cfg, 'image_size': 256, 'num_channels': 128, 'num_head_channels': 64})
    keys_small = '\n'.join(f'{k} {tuple(v.shape)}' for k, v in net.state_dict().items())
    keys_full = '\n'.join(f'{k} {tuple(v.shape)}' for k, v in full.state_dict().items())
    save('g6_unet_64.npz', x=np32(x), t=np32(t), out=np32(out), gout=np32(gout), gx=np32(gx), weight_seed=np.array(606), weight_scale=np.array(0.05),
         keys_small_sha256=np.array(hashlib.sha256(keys_small.encode()).hexdigest()),
         keys_ffhq_sha256=np.array(hashlib.sha256(keys_full.encode()).hexdigest()),
         n_params_ffhq=np.array(sum(v.numel() for v in full.state_dict().values())))


if __name__ == '__main__' and 'g6' in (sys.argv[1:] or ['g6']):
    g6_unet(import_reference() if 'main_sampling' not in sys.modules else sys.modules['main_sampling'])
