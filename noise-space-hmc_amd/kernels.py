"""Typed torch-tensor front ends of the C ABI (include/nhmc.h).

torch is plumbing here: it owns device memory and the stream; every function below checks its
operands (device, dtype, contiguity, shape) on the host, then hands raw pointers to libnhmc.so on
the CURRENT torch stream.  No function allocates inside the library; outputs are torch tensors
created here.  Nothing falls back to torch arithmetic.
"""
import ctypes as C

import torch

from . import _lib

LF_FIRST, LF_MID, LF_LAST = 0, 1, 2


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=None, name='tensor'):
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise _lib.NhmcError(f'{name} must live on the GPU (got {t.device}); libnhmc has no CPU path')
    if dtype is not None and t.dtype != dtype:
        raise _lib.NhmcError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.NhmcError(f'{name} must be contiguous')
    return C.c_void_p(t.data_ptr())


def _chains_elems(x):
    return x.shape[0], x[0].numel()


def _f64(v, n, device):
    """per-chain fp64 scalars: python float / sequence / tensor -> device double[n]"""
    if isinstance(v, torch.Tensor):
        if v.dtype != torch.float64 or v.numel() != n:
            raise _lib.NhmcError('per-chain scalar tensors must be float64 of length n_chains')
        return v
    return torch.full((n,), float(v), dtype=torch.float64, device=device) if not hasattr(v, '__len__') \
        else torch.tensor(list(v), dtype=torch.float64, device=device)


# ---- a1-a4 ----------------------------------------------------------------------------------
def leapfrog_tiles(n_elem):
    return _lib.load().nhmc_leapfrog_tiles(n_elem)


def leapfrog_ws(n_chains, n_elem, device):
    return torch.empty(n_chains * leapfrog_tiles(n_elem) * 2, dtype=torch.float64, device=device)


def leapfrog_fused(mode, x, p, g, eps, sigma_y, m_inv, sums_ws=None, g2=None):
    """In place on x, p.  eps / sigma_y: float64 device tensors [B] (or python floats)."""
    lib = _lib.load()
    B, N = _chains_elems(x)
    if p.shape != x.shape or g.shape != x.shape or (g2 is not None and g2.shape != x.shape):
        raise _lib.NhmcError('x, p, g (and g2) must have the same shape')
    eps, sigma_y = _f64(eps, B, x.device), _f64(sigma_y, B, x.device)
    if mode != LF_MID and sums_ws is None:
        raise _lib.NhmcError('FIRST/LAST modes need sums_ws')
    if sums_ws is not None and sums_ws.numel() < B * leapfrog_tiles(N) * 2:
        raise _lib.NhmcError('sums_ws too small')
    rc = lib.nhmc_leapfrog_fused(mode, _p(x, torch.float32, 'x'), _p(p, torch.float32, 'p'),
                                 _p(g, torch.float32, 'g'), _p(g2, torch.float32, 'g2'),
                                 _p(eps, torch.float64, 'eps'), _p(sigma_y, torch.float64, 'sigma_y'),
                                 float(m_inv), B, N, _p(sums_ws, torch.float64, 'sums_ws'), _stream())
    _lib.check(rc, 'nhmc_leapfrog_fused')


def leapfrog_first(x_in, x_out, p, g, eps, sigma_y, m_inv, sums_ws, g2=None):
    """FIRST mode out of place in x: x_out <- proposal position, x_in untouched, p updated in place."""
    lib = _lib.load()
    B, N = _chains_elems(x_in)
    if x_out.shape != x_in.shape or p.shape != x_in.shape or g.shape != x_in.shape or (g2 is not None and g2.shape != x_in.shape):
        raise _lib.NhmcError('x_in, x_out, p, g (and g2) must have the same shape')
    if x_out.data_ptr() == x_in.data_ptr():
        raise _lib.NhmcError('leapfrog_first is out of place: x_out must not alias x_in')
    eps, sigma_y = _f64(eps, B, x_in.device), _f64(sigma_y, B, x_in.device)
    if sums_ws is None or sums_ws.numel() < B * leapfrog_tiles(N) * 2:
        raise _lib.NhmcError('sums_ws missing or too small')
    rc = lib.nhmc_leapfrog_first(_p(x_in, torch.float32, 'x_in'), _p(x_out, torch.float32, 'x_out'), _p(p, torch.float32, 'p'),
                                 _p(g, torch.float32, 'g'), _p(g2, torch.float32, 'g2'), _p(eps, torch.float64, 'eps'),
                                 _p(sigma_y, torch.float64, 'sigma_y'), float(m_inv), B, N,
                                 _p(sums_ws, torch.float64, 'sums_ws'), _stream())
    _lib.check(rc, 'nhmc_leapfrog_first')


# ---- gradient cache (nhmc.h "Gradient cache") -------------------------------------------------
def _cache_args(g_pair, loss_pair, sel, B, N):
    """g_pair: a [2, B_total, ...] float32 tensor or a [:, lo:hi] view of one; loss_pair [2, B_total] float64 likewise;
    sel int32 [B].  -> (pair_stride, loss_stride) in elements."""
    if g_pair.dim() < 2 or g_pair.shape[0] != 2 or g_pair.shape[1] != B or g_pair[0, 0].numel() != N:
        raise _lib.NhmcError(f'g_pair must be [2, {B}, ...] with {N} elements per chain, got {tuple(g_pair.shape)}')
    if not g_pair[0].is_contiguous() or g_pair.dtype != torch.float32 or not g_pair.is_cuda:
        raise _lib.NhmcError('g_pair: float32 on the GPU, each slot contiguous')
    if loss_pair.shape != (2, B) or loss_pair.dtype != torch.float64 or loss_pair.stride(1) != 1:
        raise _lib.NhmcError('loss_pair must be float64 [2, n_chains] (or a [:, lo:hi] view)')
    if sel.dtype != torch.int32 or sel.numel() != B or not sel.is_contiguous():
        raise _lib.NhmcError('sel must be int32 [n_chains]')
    return g_pair.stride(0), loss_pair.stride(0)


def grad_cache_store(g, g2, loss, g_pair, loss_pair, sel, flip=0):
    """slot (sel ^ flip) of every chain <- (g + g2, loss)"""
    lib = _lib.load()
    B, N = _chains_elems(g)
    ps, ls = _cache_args(g_pair, loss_pair, sel, B, N)
    if g2 is not None and g2.shape != g.shape:
        raise _lib.NhmcError('g and g2 must have the same shape')
    rc = lib.nhmc_grad_cache_store(_p(g, torch.float32, 'g'), _p(g2, torch.float32, 'g2'), _p(loss, torch.float64, 'loss'),
                                   C.c_void_p(g_pair.data_ptr()), C.c_void_p(loss_pair.data_ptr()), _p(sel, torch.int32),
                                   int(flip), ps, ls, B, N, _stream())
    _lib.check(rc, 'nhmc_grad_cache_store')


def leapfrog_first_cached(x_in, x_out, p, g_pair, sel, eps, sigma_y, m_inv, sums_ws):
    """FIRST half step (out of place in x) with the gradient read from the cache slot sel[chain]."""
    lib = _lib.load()
    B, N = _chains_elems(x_in)
    if x_out.shape != x_in.shape or p.shape != x_in.shape:
        raise _lib.NhmcError('x_in, x_out, p must have the same shape')
    if g_pair.dim() < 2 or g_pair.shape[0] != 2 or g_pair.shape[1] != B or g_pair[0, 0].numel() != N or \
            not g_pair[0].is_contiguous() or sel.numel() != B:
        raise _lib.NhmcError('g_pair must be [2, n_chains, ...] like x, sel int32 [n_chains]')
    eps, sigma_y = _f64(eps, B, x_in.device), _f64(sigma_y, B, x_in.device)
    if sums_ws is None or sums_ws.numel() < B * leapfrog_tiles(N) * 2:
        raise _lib.NhmcError('sums_ws missing or too small')
    rc = lib.nhmc_leapfrog_first_cached(_p(x_in, torch.float32, 'x_in'), _p(x_out, torch.float32, 'x_out'),
                                        _p(p, torch.float32, 'p'), C.c_void_p(g_pair.data_ptr()), _p(sel, torch.int32, 'sel'),
                                        g_pair.stride(0), _p(eps, torch.float64), _p(sigma_y, torch.float64), float(m_inv),
                                        B, N, _p(sums_ws, torch.float64), _stream())
    _lib.check(rc, 'nhmc_leapfrog_first_cached')


def leapfrog_last_cached(x, p, g, g2, g_pair, loss_pair, sel, loss, eps, sigma_y, m_inv, sums_ws):
    """LAST step; slot 1 - sel[chain] <- (g + g2, loss) of the trajectory's end point."""
    lib = _lib.load()
    B, N = _chains_elems(x)
    if p.shape != x.shape or g.shape != x.shape or (g2 is not None and g2.shape != x.shape):
        raise _lib.NhmcError('x, p, g (and g2) must have the same shape')
    ps, ls = _cache_args(g_pair, loss_pair, sel, B, N)
    eps, sigma_y = _f64(eps, B, x.device), _f64(sigma_y, B, x.device)
    if sums_ws is None or sums_ws.numel() < B * leapfrog_tiles(N) * 2:
        raise _lib.NhmcError('sums_ws missing or too small')
    if loss.numel() != B:
        raise _lib.NhmcError('loss must have one entry per chain')
    rc = lib.nhmc_leapfrog_last_cached(_p(x, torch.float32, 'x'), _p(p, torch.float32, 'p'), _p(g, torch.float32, 'g'),
                                       _p(g2, torch.float32, 'g2'), C.c_void_p(g_pair.data_ptr()), _p(sel, torch.int32),
                                       ps, _p(loss, torch.float64, 'loss'), C.c_void_p(loss_pair.data_ptr()), ls,
                                       _p(eps, torch.float64), _p(sigma_y, torch.float64), float(m_inv), B, N,
                                       _p(sums_ws, torch.float64), _stream())
    _lib.check(rc, 'nhmc_leapfrog_last_cached')


def grad_cache_flip(accept, sel):
    lib = _lib.load()
    _lib.check(lib.nhmc_grad_cache_flip(_p(accept, torch.int32, 'accept'), _p(sel, torch.int32, 'sel'), sel.numel(), _stream()),
               'nhmc_grad_cache_flip')


# ---- a9-a11 ---------------------------------------------------------------------------------
def _mix_shapes(xt, e):
    B, Cc = xt.shape[0], xt.shape[1]
    hw = xt[0, 0].numel()
    if e.shape[0] != B or e.shape[2:] != xt.shape[2:]:
        raise _lib.NhmcError(f'score output shape {tuple(e.shape)} does not match xt {tuple(xt.shape)}')
    return B, Cc, hw, e.shape[1]


def _alpha(a, B, device):
    if a.dim() == 1 and a.numel() == B and a.dtype == torch.float32 and a.device == device and a.is_contiguous():
        return a                                     # already the per-chain array the kernels take (engine's cached tables)
    a = a.reshape(-1)
    if a.numel() == 1 and B > 1:
        a = a.expand(B)
    if a.numel() != B:
        raise _lib.NhmcError('alpha-bar must have one entry per chain')
    return a.to(device=device, dtype=torch.float32).contiguous()


def ddim_mix_fwd(xt, e, at, at_next, final_clip=False, want=('xt_next',), out=None):
    """Returns a dict with the requested outputs among 'xt_next', 'x0_t', 'add_up'.  out: tensor to write xt_next into."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    if out is not None and (out.shape != xt.shape or 'xt_next' not in want):
        raise _lib.NhmcError('ddim_mix_fwd: out must have the shape of xt and xt_next must be wanted')
    out = {k: (out if (k == 'xt_next' and out is not None) else torch.empty_like(xt)) for k in want}
    rc = lib.nhmc_ddim_mix_fwd(_p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                               int(final_clip), _p(out.get('xt_next')), _p(out.get('x0_t')), _p(out.get('add_up')),
                               B, Cc, hw, _stream())
    _lib.check(rc, 'nhmc_ddim_mix_fwd')
    return out


def ddim_map_back(x0_t, add_up, at_next):
    lib = _lib.load()
    B, N = _chains_elems(x0_t)
    out = torch.empty_like(x0_t)
    an = _alpha(at_next, B, x0_t.device)                 # keep temporaries alive until after the launch call:
    rc = lib.nhmc_ddim_map_back(_p(x0_t, torch.float32, 'x0_t'), _p(add_up, torch.float32, 'add_up'),   # a freed block
                                _p(an), _p(out), B, N, _stream())                                          # can be re-issued
    _lib.check(rc, 'nhmc_ddim_map_back')
    return out


def ddim_mix_bwd(gout, xt, e, at, at_next, final_clip=False, gout2=None, g_x0=None, g_e_out=None, want_g_e=True):
    """-> (g_xt [B,C,H,W], g_e [B,e_channels,H,W]).  g_x0: split form (gout is then d/d add_up).
    g_e_out: a persistent buffer whose sigma-channels are already zero (they are then not rewritten).
    want_g_e=False: the score carries no gradient; g_e is not formed (returned as None)."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    g_xt = torch.empty_like(xt)
    g_e = None if not want_g_e else (g_e_out if g_e_out is not None else torch.empty_like(e))
    if g_e is not None and g_e.shape != e.shape:
        raise _lib.NhmcError('g_e_out must have the shape of the score output')
    rc = lib.nhmc_ddim_mix_bwd(_p(gout, torch.float32, 'gout'), _p(gout2, torch.float32, 'gout2'),
                               _p(g_x0, torch.float32, 'g_x0'), _p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                               int(final_clip), _p(g_xt), _p(g_e), int(g_e_out is None), B, Cc, hw, _stream())
    _lib.check(rc, 'nhmc_ddim_mix_bwd')
    return g_xt, g_e


def ddim_mix_bwd_inpaint(xt, e, at, at_next, y, slot, g_e_out=None, loss_out=None):
    """Last-step VJP fused with the inpainting data term -> (loss [B] float64, g_xt, g_e)."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.empty_like(e)
    tiles = leapfrog_tiles(Cc * hw)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    rc = lib.nhmc_ddim_mix_bwd_inpaint(_p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                                       _p(y, torch.float32, 'y'), _p(slot, torch.int32, 'slot'), y.shape[1], _p(g_xt),
                                       _p(g_e), int(g_e_out is None), _p(ws), B, Cc, hw, _stream())
    _lib.check(rc, 'nhmc_ddim_mix_bwd_inpaint')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


def ddim_mix_bwd_inpaint_px(xt, e, at, at_next, y, mask_words, prefix, g_e_out=None, loss_out=None):
    """Whole-pixel-mask form of ddim_mix_bwd_inpaint: bit mask + prefix counts instead of the dense slot map."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.empty_like(e)
    tiles = lib.nhmc_inpaint_px_tiles(Cc, hw)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    rc = lib.nhmc_ddim_mix_bwd_inpaint_px(_p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                                          _p(y, torch.float32, 'y'), _p(mask_words, torch.int32, 'mask_words'),
                                          _p(prefix, torch.int32, 'prefix'), y.shape[1], _p(g_xt), _p(g_e),
                                          int(g_e_out is None), _p(ws), B, Cc, hw, _stream())
    _lib.check(rc, 'nhmc_ddim_mix_bwd_inpaint_px')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


def ddim_mix_bwd_sr(xt, e, at, at_next, y, ratio, g_e_out=None, loss_out=None):
    """Last-step VJP fused with the super-resolution data term -> (loss [B] float64, g_xt, g_e)."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    dim = xt.shape[2]
    if xt.shape[3] != dim:
        raise _lib.NhmcError('square images only')
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.zeros_like(e)          # the kernel writes channels [0, C) only
    tiles = lib.nhmc_sr_vjp_tiles(Cc, dim, ratio)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    rc = lib.nhmc_ddim_mix_bwd_sr(_p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                                  _p(y, torch.float32, 'y'), ratio, _p(g_xt), _p(g_e), _p(ws), B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_ddim_mix_bwd_sr')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


# ---- a12-a15 --------------------------------------------------------------------------------
NO_LOSS = object()      # pass as loss_out: the per-chain loss is not wanted (a MID leapfrog step uses only the gradient)


def sum_partials(ws, tiles, n_chains, stride=1, offset=0, out=None):
    """Second pass of the two-pass loss: the tile partials of each chain added in a fixed order.  out=NO_LOSS skips the
    launch and returns None -- the sampler reads the loss of a trajectory's first and last evaluation only
    (main_sampling.py:697,717), so the L - 1 evaluations in between need the gradient alone."""
    if out is NO_LOSS:
        return None
    lib = _lib.load()
    if out is None:
        out = torch.empty(n_chains, dtype=torch.float64, device=ws.device)
    elif out.numel() != n_chains or out.dtype != torch.float64:
        raise _lib.NhmcError('sum_partials: out must be float64 of length n_chains')
    _lib.check(lib.nhmc_sum_partials(_p(ws, torch.float64), tiles, stride, offset, n_chains, _p(out), _stream()),
               'nhmc_sum_partials')
    return out


def data_inpaint(xt, y, slot, apply_clip=True, loss_out=None):
    """-> (loss [B] float64, g_xt).  slot: int32 [N] CHW map into y (-1 = masked)."""
    lib = _lib.load()
    B, N = _chains_elems(xt)
    M = y.shape[1]
    if slot.numel() != N or y.shape[0] != B:
        raise _lib.NhmcError('slot / y shape mismatch')
    tiles = lib.nhmc_data_tiles(N)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    g = torch.empty_like(xt)
    rc = lib.nhmc_data_inpaint(_p(xt, torch.float32, 'xt'), _p(y, torch.float32, 'y'), _p(slot, torch.int32, 'slot'),
                               int(apply_clip), _p(g), _p(ws), B, N, M, _stream())
    _lib.check(rc, 'nhmc_data_inpaint')
    return sum_partials(ws, tiles, B, out=loss_out), g


def inpaint_H(x, kept_chw):
    lib = _lib.load()
    B, N = _chains_elems(x)
    M = kept_chw.numel()
    y = torch.empty(B, M, dtype=torch.float32, device=x.device)
    _lib.check(lib.nhmc_inpaint_H(_p(x, torch.float32, 'x'), _p(kept_chw, torch.int32), _p(y), B, N, M, _stream()),
               'nhmc_inpaint_H')
    return y


def inpaint_Ht(y, slot, n_elem):
    lib = _lib.load()
    B, M = y.shape
    x = torch.empty(B, n_elem, dtype=torch.float32, device=y.device)
    _lib.check(lib.nhmc_inpaint_Ht(_p(y, torch.float32, 'y'), _p(slot, torch.int32), _p(x), B, n_elem, M, _stream()),
               'nhmc_inpaint_Ht')
    return x


def data_sr(xt, y, ratio, apply_clip=True, loss_out=None):
    lib = _lib.load()
    B, Cc, dim = xt.shape[0], xt.shape[1], xt.shape[2]
    if xt.shape[3] != dim:
        raise _lib.NhmcError('square images only')
    tiles = lib.nhmc_sr_tiles(Cc, dim, ratio)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    g = torch.empty_like(xt)
    rc = lib.nhmc_data_sr(_p(xt, torch.float32, 'xt'), _p(y, torch.float32, 'y'), ratio, int(apply_clip), _p(g),
                          _p(ws), B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_data_sr')
    return sum_partials(ws, tiles, B, out=loss_out), g


def sr_H(x, ratio):
    lib = _lib.load()
    B, Cc, dim = x.shape[0], x.shape[1], x.shape[2]
    y = torch.empty(B, Cc * (dim // ratio) ** 2, dtype=torch.float32, device=x.device)
    _lib.check(lib.nhmc_sr_H(_p(x, torch.float32, 'x'), _p(y), ratio, B, Cc, dim, _stream()), 'nhmc_sr_H')
    return y


def sr_Ht(y, ratio, channels, dim, scale):
    lib = _lib.load()
    B = y.shape[0]
    x = torch.empty(B, channels * dim * dim, dtype=torch.float32, device=y.device)
    _lib.check(lib.nhmc_sr_Ht(_p(y, torch.float32, 'y'), _p(x), ratio, float(scale), B, channels, dim, _stream()),
               'nhmc_sr_Ht')
    return x


def _host_w(w, channels):
    """[V[0,0] .. V[C-1,0], s, U[0,0]]: the kernels read channels + 2 host floats (ABI 2; ABI 1 took `channels`)."""
    if len(w) != channels + 2:
        raise _lib.NhmcError(f'colorization weights: channels + 2 = {channels + 2} values (V[:,0], s, U[0,0]), got {len(w)}')
    arr = (C.c_float * len(w))(*[float(v) for v in w])
    return C.cast(arr, C.c_void_p), arr


def data_color(xt, y, w, apply_clip=True, loss_out=None):
    """w: host sequence [V[0,0] .. V[C-1,0], s, U[0,0]] of the grey row's SVD -> (loss [B] float64, g_xt)"""
    lib = _lib.load()
    B, Cc, hw = xt.shape[0], xt.shape[1], xt[0, 0].numel()
    tiles = lib.nhmc_color_tiles(hw)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    g = torch.empty_like(xt)
    wp, keep = _host_w(w, Cc)
    rc = lib.nhmc_data_color(_p(xt, torch.float32, 'xt'), _p(y, torch.float32, 'y'), wp, int(apply_clip), _p(g), _p(ws),
                             B, Cc, hw, _stream())
    _lib.check(rc, 'nhmc_data_color')
    return sum_partials(ws, tiles, B, out=loss_out), g


def ddim_mix_bwd_color(xt, e, at, at_next, y, w, g_e_out=None, loss_out=None):
    """Last-step VJP fused with the colorization data term -> (loss [B] float64, g_xt, g_e)."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    tiles = lib.nhmc_color_tiles(hw)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.zeros_like(e)
    wp, keep = _host_w(w, Cc)
    rc = lib.nhmc_ddim_mix_bwd_color(_p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                                     _p(y, torch.float32, 'y'), wp, _p(g_xt), _p(g_e), _p(ws), B, Cc, hw, _stream())
    _lib.check(rc, 'nhmc_ddim_mix_bwd_color')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


def color_H(x, w):
    lib = _lib.load()
    B, Cc, hw = x.shape[0], x.shape[1], x[0, 0].numel()
    y = torch.empty(B, hw, dtype=torch.float32, device=x.device)
    wp, keep = _host_w(w, Cc)
    _lib.check(lib.nhmc_color_H(_p(x, torch.float32, 'x'), wp, _p(y), B, Cc, hw, _stream()), 'nhmc_color_H')
    return y


def color_Ht(y, w, channels, pinv=False):
    lib = _lib.load()
    B, hw = y.shape
    x = torch.empty(B, channels * hw, dtype=torch.float32, device=y.device)
    wp, keep = _host_w(w, channels)
    _lib.check(lib.nhmc_color_Ht(_p(y, torch.float32, 'y'), wp, int(pinv), _p(x), B, channels, hw, _stream()), 'nhmc_color_Ht')
    return x


def cs_H(x, kslot, m):
    lib = _lib.load()
    B, Cc, dim = x.shape[0], x.shape[1], x.shape[2]
    y = torch.zeros(B, m, dtype=torch.float32, device=x.device)
    tmp = torch.empty_like(x)
    _lib.check(lib.nhmc_cs_H(_p(x, torch.float32, 'x'), _p(kslot, torch.int32), _p(y), _p(tmp), B, Cc, dim, m, _stream()),
               'nhmc_cs_H')
    return y


def cs_Ht(y, kslot, channels, dim):
    lib = _lib.load()
    B, m = y.shape
    x = torch.empty(B, channels, dim, dim, dtype=torch.float32, device=y.device)
    tmp = torch.empty_like(x)
    _lib.check(lib.nhmc_cs_Ht(_p(y, torch.float32, 'y'), _p(kslot, torch.int32), _p(x), _p(tmp), B, channels, dim, m,
                              _stream()), 'nhmc_cs_Ht')
    return x.reshape(B, -1)


def _cs_tmp(xt, dim):
    return torch.empty((1 if dim == 256 else 2,) + tuple(xt.shape), dtype=torch.float32, device=xt.device)


def data_cs(xt, y_spec, apply_clip=True, loss_out=None):
    """y_spec: the observation in spectrum layout [B, C, d, d], NaN where not observed (nhmc.h, nhmc_data_cs)."""
    lib = _lib.load()
    B, Cc, dim = xt.shape[0], xt.shape[1], xt.shape[2]
    if y_spec.numel() != xt.numel():
        raise _lib.NhmcError('data_cs: y_spec must have one entry per image element (spectrum layout)')
    tiles = lib.nhmc_cs_tiles(Cc, dim)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    tmp = _cs_tmp(xt, dim)
    g = torch.empty_like(xt)
    rc = lib.nhmc_data_cs(_p(xt, torch.float32, 'xt'), _p(y_spec, torch.float32, 'y_spec'), int(apply_clip),
                          _p(g), _p(ws), _p(tmp), B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_data_cs')
    return sum_partials(ws, tiles, B, out=loss_out), g


def data_cs_vjp(xt_next, y_spec, xt, e, at, at_next, g_e_out=None, loss_out=None):
    """Walsh-Hadamard CS data term on the clipped decode + VJP of the last DDIM step in the last row pass
    -> (loss [B] float64, g_xt, g_e).  y_spec: as in data_cs."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    dim = xt.shape[2]
    if y_spec.numel() != xt.numel():
        raise _lib.NhmcError('data_cs_vjp: y_spec must have one entry per image element (spectrum layout)')
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    tiles = lib.nhmc_cs_tiles(Cc, dim)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    tmp = _cs_tmp(xt, dim)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.zeros_like(e)
    rc = lib.nhmc_data_cs_vjp(_p(xt_next, torch.float32, 'xt_next'), _p(y_spec, torch.float32, 'y_spec'),
                              _p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next), _p(g_xt),
                              _p(g_e), _p(ws), _p(tmp), B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_data_cs_vjp')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


def spectral_apply(x, L, R, Dmap, LoT, RoT):
    """out_c = Lo (D_c o (L^T X_c R)) Ro^T ; x: [B,C,d,d]"""
    lib = _lib.load()
    B, Cc, dim = x.shape[0], x.shape[1], x.shape[2]
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    rc = lib.nhmc_spectral_apply(_p(x, torch.float32, 'x'), _p(L, torch.float32), _p(R, torch.float32),
                                 _p(Dmap, torch.float32), _p(LoT, torch.float32), _p(RoT, torch.float32), _p(out),
                                 _p(tmp), B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_spectral_apply')
    return out


def sandwich_rect(x, S1, S2, mul=None):
    """x: [n_img, K1, R1]; t = x^T S1; out = (t^T S2) o mul -> [n_img, C1, C2] = (S1^T x S2) o mul  (mul [C1, C2] or None)"""
    lib = _lib.load()
    n, K1, R1 = x.shape
    C1, C2 = S1.shape[1], S2.shape[1]
    if S1.shape[0] != K1 or S2.shape[0] != R1:
        raise _lib.NhmcError('sandwich_rect: factor shapes do not chain')
    if mul is not None and tuple(mul.shape) != (C1, C2):
        raise _lib.NhmcError('sandwich_rect: the multiplier map must be [C1, C2]')
    out = torch.empty(n, C1, C2, dtype=torch.float32, device=x.device)
    tmp = torch.empty(n, R1, C1, dtype=torch.float32, device=x.device)
    rc = lib.nhmc_sandwich_rect(_p(x, torch.float32, 'x'), _p(S1, torch.float32), _p(S2, torch.float32), _p(mul, torch.float32, 'mul'),
                                _p(out), _p(tmp), n, K1, R1, C1, C2, _stream())
    _lib.check(rc, 'nhmc_sandwich_rect')
    return out


def _srconv_factors(f, dim):
    """f = (V1 [d, sd], V1T [sd, d], U [sd, sd], UT [sd, sd], S [sd, sd]) -> sd, after a shape check"""
    V1, V1T, U, UT, S = f
    sd = U.shape[0]
    if tuple(V1.shape) != (dim, sd) or tuple(V1T.shape) != (sd, dim) or tuple(U.shape) != (sd, sd) or \
            tuple(UT.shape) != (sd, sd) or tuple(S.shape) != (sd, sd):
        raise _lib.NhmcError('SRConv factors: V1 [d, sd], V1^T [sd, d], U, U^T, S [sd, sd]')
    return sd


def data_srconv(xt, y, factors, apply_clip=True, loss_out=None):
    """factors: (V1, V1T, U, UT, S) as nhmc.operators.SRConv keeps them -> (loss [B] float64, g_xt)"""
    lib = _lib.load()
    B, Cc, dim = xt.shape[0], xt.shape[1], xt.shape[2]
    sd = _srconv_factors(factors, dim)
    tiles = lib.nhmc_srconv_tiles(Cc, sd)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    tmp = torch.empty(B * Cc * (dim * sd + 3 * sd * sd), dtype=torch.float32, device=xt.device)
    g = torch.empty_like(xt)
    rc = lib.nhmc_data_srconv(_p(xt, torch.float32, 'xt'), _p(y, torch.float32, 'y'), *[_p(t, torch.float32) for t in factors],
                              int(apply_clip), _p(g), _p(ws), _p(tmp), B, Cc, dim, sd, _stream())
    _lib.check(rc, 'nhmc_data_srconv')
    return sum_partials(ws, tiles, B, out=loss_out), g


def data_srconv_vjp(xt_next, y, factors, xt, e, at, at_next, g_e_out=None, loss_out=None):
    """Bicubic / strided-convolution data term on the clipped decode + VJP of the last DDIM step in the last product's
    epilogue -> (loss [B] float64, g_xt, g_e)."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    dim = xt.shape[2]
    sd = _srconv_factors(factors, dim)
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    tiles = lib.nhmc_srconv_tiles(Cc, sd)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    tmp = torch.empty(B * Cc * (dim * sd + 3 * sd * sd), dtype=torch.float32, device=xt.device)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.zeros_like(e)
    rc = lib.nhmc_data_srconv_vjp(_p(xt_next, torch.float32, 'xt_next'), _p(y, torch.float32, 'y'),
                                  *[_p(t, torch.float32) for t in factors], _p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec,
                                  _p(at), _p(at_next), _p(g_xt), _p(g_e), _p(ws), _p(tmp), B, Cc, dim, sd, _stream())
    _lib.check(rc, 'nhmc_data_srconv_vjp')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


def spectral_project(y, L, R):
    """y^ = L^T Y R per channel image (L = U1, R = U2): the observation in the operator's left singular basis."""
    lib = _lib.load()
    B, Cc, dim = y.shape[0], y.shape[1], y.shape[2]
    out, tmp = torch.empty_like(y), torch.empty_like(y)
    rc = lib.nhmc_spectral_project(_p(y, torch.float32, 'y'), _p(L, torch.float32), _p(R, torch.float32), _p(out), _p(tmp),
                                   B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_spectral_project')
    return out


def data_spectral(xt, y, factors, Dmap, apply_clip=True, loss_out=None, projected=False, DmapT=None):
    """factors: packed [8,d,d] = U1,U2,V1,V2,U1^T,U2^T,V1^T,V2^T -> (loss [B] float64, g_xt).
    Eight-product (reference-order) form: `y` is the observation with every channel plane TRANSPOSED and DmapT the
    multiplier map likewise (nhmc.h, nhmc_data_spectral).  projected: `y` is spectral_project(y, U1, U2) (natural
    orientation) and the four-product form runs."""
    lib = _lib.load()
    B, Cc, dim = xt.shape[0], xt.shape[1], xt.shape[2]
    tiles = lib.nhmc_spectral_tiles(Cc, dim)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    tmp = torch.empty((1 if projected else 2,) + tuple(xt.shape), dtype=torch.float32, device=xt.device)
    g = torch.empty_like(xt)
    if projected:
        rc = lib.nhmc_data_spectral_proj(_p(xt, torch.float32, 'xt'), _p(y, torch.float32, 'y'), _p(factors, torch.float32, 'factors'),
                                         _p(Dmap, torch.float32), int(apply_clip), _p(g), _p(ws), _p(tmp), B, Cc, dim, _stream())
    else:
        if DmapT is None:
            raise _lib.NhmcError('data_spectral: the eight-product form needs DmapT (Dmap with every plane transposed)')
        rc = lib.nhmc_data_spectral(_p(xt, torch.float32, 'xt'), _p(y, torch.float32, 'yT'), _p(factors, torch.float32, 'factors'),
                                    _p(Dmap, torch.float32), _p(DmapT, torch.float32, 'DmapT'), int(apply_clip), _p(g), _p(ws), _p(tmp),
                                    B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_data_spectral')
    return sum_partials(ws, tiles, B, out=loss_out), g


def data_spectral_vjp(xt_next, y, factors, Dmap, xt, e, at, at_next, g_e_out=None, loss_out=None, projected=False, DmapT=None):
    """Spectral data term on the clipped decode `xt_next` + VJP of the last DDIM step (inputs xt, e) in the last
    product's epilogue -> (loss [B] float64, g_xt, g_e).  y / DmapT / projected: as in data_spectral."""
    lib = _lib.load()
    B, Cc, hw, ec = _mix_shapes(xt, e)
    dim = xt.shape[2]
    at, at_next = _alpha(at, B, xt.device), _alpha(at_next, B, xt.device)
    tiles = lib.nhmc_spectral_tiles(Cc, dim)
    ws = torch.empty(B * tiles, dtype=torch.float64, device=xt.device)
    tmp = torch.empty((1 if projected else 2,) + tuple(xt.shape), dtype=torch.float32, device=xt.device)
    g_xt = torch.empty_like(xt)
    g_e = g_e_out if g_e_out is not None else torch.zeros_like(e)          # channels [0, C) are written
    if projected:
        rc = lib.nhmc_data_spectral_proj_vjp(_p(xt_next, torch.float32, 'xt_next'), _p(y, torch.float32, 'y'),
                                             _p(factors, torch.float32, 'factors'), _p(Dmap, torch.float32),
                                             _p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                                             _p(g_xt), _p(g_e), _p(ws), _p(tmp), B, Cc, dim, _stream())
    else:
        if DmapT is None:
            raise _lib.NhmcError('data_spectral_vjp: the eight-product form needs DmapT (Dmap with every plane transposed)')
        rc = lib.nhmc_data_spectral_vjp(_p(xt_next, torch.float32, 'xt_next'), _p(y, torch.float32, 'yT'),
                                        _p(factors, torch.float32, 'factors'), _p(Dmap, torch.float32), _p(DmapT, torch.float32, 'DmapT'),
                                        _p(xt, torch.float32, 'xt'), _p(e, torch.float32, 'e'), ec, _p(at), _p(at_next),
                                        _p(g_xt), _p(g_e), _p(ws), _p(tmp), B, Cc, dim, _stream())
    _lib.check(rc, 'nhmc_data_spectral_vjp')
    return sum_partials(ws, tiles, B, out=loss_out), g_xt, g_e


# ---- a5-a7 ----------------------------------------------------------------------------------
def hamiltonian(sums_ws, n_elem, loss, sigma_y, m_inv, want_terms=False, sel=None):
    """sel (int32 [B]): loss is then the gradient cache's loss_pair [2, B] and chain c uses loss[sel[c], c]."""
    lib = _lib.load()
    if sel is not None:
        B = sel.numel()
        if loss.shape != (2, B) or loss.dtype != torch.float64 or loss.stride(1) != 1:
            raise _lib.NhmcError('hamiltonian(sel=...): loss must be the float64 [2, n_chains] cache')
        H = torch.empty(B, dtype=torch.float32, device=loss.device)
        terms = torch.empty(B, 3, dtype=torch.float64, device=loss.device) if want_terms else None
        sy = _f64(sigma_y, B, loss.device)
        rc = lib.nhmc_hamiltonian_cached(_p(sums_ws, torch.float64), leapfrog_tiles(n_elem), C.c_void_p(loss.data_ptr()),
                                         _p(sel, torch.int32, 'sel'), loss.stride(0), _p(sy), float(m_inv), _p(H), _p(terms),
                                         B, _stream())
        _lib.check(rc, 'nhmc_hamiltonian_cached')
        return (H, terms) if want_terms else H
    B = loss.numel()
    H = torch.empty(B, dtype=torch.float32, device=loss.device)
    terms = torch.empty(B, 3, dtype=torch.float64, device=loss.device) if want_terms else None
    sy = _f64(sigma_y, B, loss.device)
    rc = lib.nhmc_hamiltonian(_p(sums_ws, torch.float64), leapfrog_tiles(n_elem), _p(loss, torch.float64, 'loss'),
                              _p(sy), float(m_inv), _p(H), _p(terms), B, _stream())
    _lib.check(rc, 'nhmc_hamiltonian')
    return (H, terms) if want_terms else H


def metropolis(H0, H1, u, active=None):
    lib = _lib.load()
    B = H0.numel()
    acc = torch.empty(B, dtype=torch.int32, device=H0.device)
    dH = torch.empty(B, dtype=torch.float32, device=H0.device)
    rc = lib.nhmc_metropolis(_p(H0, torch.float32), _p(H1, torch.float32), _p(u, torch.float32, 'u'),
                             _p(active, torch.int32), _p(acc), _p(dH), B, _stream())
    _lib.check(rc, 'nhmc_metropolis')
    return acc, dH


def schedule_begin(state, sigma_0, epochs, sampling):
    lib = _lib.load()
    B = state['epoch'].numel()
    rc = lib.nhmc_schedule_begin(_p(state['epoch'], torch.int32), _p(state['tau'], torch.float64),
                                 _p(state['eps'], torch.float64), _p(state['sigma_y'], torch.float64),
                                 _p(state['eps_eff'], torch.float64), _p(state['active'], torch.int32),
                                 float(sigma_0), epochs, sampling, B, _stream())
    _lib.check(rc, 'nhmc_schedule_begin')


def accept_commit(accept, epoch, x, x_prop, xt_prop, samples, epochs, sampling):
    lib = _lib.load()
    B, N = _chains_elems(x)
    rc = lib.nhmc_accept_commit(_p(accept, torch.int32), _p(epoch, torch.int32), _p(x, torch.float32, 'x'),
                                _p(x_prop, torch.float32, 'x_prop'), _p(xt_prop, torch.float32, 'xt_prop'),
                                _p(samples, torch.float32, 'samples'), epochs, sampling, B, N, _stream())
    _lib.check(rc, 'nhmc_accept_commit')


def schedule_end(accept, state):
    lib = _lib.load()
    B = accept.numel()
    rc = lib.nhmc_schedule_end(_p(accept, torch.int32), _p(state['active'], torch.int32),
                               _p(state['epoch'], torch.int32), _p(state['rejected'], torch.int32),
                               _p(state['tau'], torch.float64), _p(state['eps'], torch.float64),
                               _p(state.get('n_accept'), torch.int32), _p(state.get('n_reject'), torch.int32),
                               B, _stream())
    _lib.check(rc, 'nhmc_schedule_end')


def latent_commit(accept, st, final_phase, keep, x, x_prop, xt_last, xt_prop, samples):
    lib = _lib.load()
    B, N = _chains_elems(x)
    rc = lib.nhmc_latent_commit(_p(accept, torch.int32), _p(st['has_prev'], torch.int32), _p(st['count'], torch.int32),
                                int(final_phase), int(keep), _p(x, torch.float32, 'x'), _p(x_prop, torch.float32, 'x_prop'),
                                _p(xt_last, torch.float32, 'xt_last'), _p(xt_prop, torch.float32, 'xt_prop'),
                                _p(samples, torch.float32, 'samples'), B, N, _stream())
    _lib.check(rc, 'nhmc_latent_commit')


def schedule_end_latent(accept, st, sigma_y_on_accept, final_phase):
    lib = _lib.load()
    rc = lib.nhmc_schedule_end_latent(_p(accept, torch.int32), _p(st['rejected'], torch.int32), _p(st['tau'], torch.float64),
                                      _p(st['eps'], torch.float64), _p(st['sigma_y'], torch.float64),
                                      _p(st['count'], torch.int32), _p(st['has_prev'], torch.int32),
                                      _p(st.get('n_accept'), torch.int32), float(sigma_y_on_accept), int(final_phase),
                                      accept.numel(), _stream())
    _lib.check(rc, 'nhmc_schedule_end_latent')


def leapfrog_mass(mode, x, p, g, inv_m, eps, sigma_y, sums_ws=None, g2=None, z=None, std_m=None, welford_on=None,
                  mean=None, m2=None, l=0):
    lib = _lib.load()
    B, N = _chains_elems(x)
    ep, sy = _f64(eps, B, x.device), _f64(sigma_y, B, x.device)          # two live tensors (never alias a freed block)
    rc = lib.nhmc_leapfrog_mass(mode, _p(x, torch.float32, 'x'), _p(p, torch.float32, 'p'), _p(z, torch.float32, 'z'),
                                _p(g, torch.float32, 'g'), _p(g2, torch.float32, 'g2'), _p(inv_m, torch.float32, 'inv_m'),
                                _p(std_m, torch.float32, 'std_m'), _p(ep), _p(sy),
                                _p(welford_on, torch.int32), _p(mean, torch.float32), _p(m2, torch.float32), int(l), B, N,
                                _p(sums_ws, torch.float64), _stream())
    _lib.check(rc, 'nhmc_leapfrog_mass')


def mass_from_variance(m2, L, flags, inv_m, std_m, tables, ws=None):
    """tables: (std_table, inv_table) float32 [N] on the device, by rank (nhmc.schedule.mass_tables(N, device))."""
    lib = _lib.load()
    B, N = _chains_elems(m2)
    std_t, inv_t = tables
    if std_t.numel() != N or inv_t.numel() != N:
        raise _lib.NhmcError('mass tables must have one entry per element (rank)')
    need = lib.nhmc_mass_sort_ws_bytes(B, N)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=m2.device)
    rc = lib.nhmc_mass_from_variance(_p(m2, torch.float32, 'm2'), int(L), _p(flags, torch.int32), _p(std_t, torch.float32, 'std_table'),
                                     _p(inv_t, torch.float32, 'inv_table'), _p(inv_m, torch.float32),
                                     _p(std_m, torch.float32), _p(ws), ws.numel(), B, N, _stream())
    _lib.check(rc, 'nhmc_mass_from_variance')
    return ws


def schedule_begin_mass(state, sigma_table, burn, epochs, sampling):
    lib = _lib.load()
    B = state['epoch'].numel()
    rc = lib.nhmc_schedule_begin_mass(_p(state['epoch'], torch.int32), _p(state['tau'], torch.float64),
                                      _p(state['eps'], torch.float64), _p(state['sigma_y'], torch.float64),
                                      _p(state['eps_eff'], torch.float64), _p(state['active'], torch.int32),
                                      _p(state['welford_on'], torch.int32), _p(sigma_table, torch.float64), burn, epochs,
                                      sampling, B, _stream())
    _lib.check(rc, 'nhmc_schedule_begin_mass')


def vq_nearest(z, codebook):
    """z [B, D, h, w], codebook [n_embed, D] -> (z_q = z + (e[k*] - z), k* int32 [B, h, w])."""
    lib = _lib.load()
    B, Dd = z.shape[0], z.shape[1]
    hw = z[0, 0].numel()
    if codebook.dim() != 2 or codebook.shape[1] != Dd:
        raise _lib.NhmcError(f'codebook {tuple(codebook.shape)} does not match latent channels {Dd}')
    zq = torch.empty_like(z)
    idx = torch.empty((B,) + tuple(z.shape[2:]), dtype=torch.int32, device=z.device)
    _lib.check(lib.nhmc_vq_nearest(_p(z, torch.float32, 'z'), _p(codebook, torch.float32, 'codebook'), _p(zq), _p(idx),
                                   B, Dd, hw, codebook.shape[0], _stream()), 'nhmc_vq_nearest')
    return zq, idx


def _gn_shape(x, gamma, groups, film, pre):
    B, Cc = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    if gamma.numel() != Cc or Cc % groups or hw % 4:
        raise _lib.NhmcError(f'fused GroupNorm: shape {tuple(x.shape)} / {groups} groups not covered (hw % 4 == 0, C % G == 0)')
    stride = pstride = 0
    if film is not None:
        if film.dim() != 2 or film.shape != (B, 2 * Cc) or film.stride(1) != 1:
            raise _lib.NhmcError('fused GroupNorm: film must be [B, 2C] (scale | shift) with unit inner stride')
        stride = film.stride(0)
    if pre is not None:
        if pre.dtype != torch.float32 or not pre.is_cuda or pre.stride(-1) != 1 or pre.shape not in ((Cc,), (B, Cc)):
            raise _lib.NhmcError('fused GroupNorm: pre must be float32 [C] or [B, C] on the GPU')
        pstride = pre.stride(0) if pre.dim() == 2 else 0
    return B, Cc, hw, stride, pstride


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def gn_act_fwd(x, gamma, beta, groups, eps, act, film=None, pre=None):
    """GroupNorm of (x + pre) (+ FiLM) (+ SiLU) forward -> (y, ws, splits); ws feeds gn_act_bwd."""
    lib = _lib.load()
    B, Cc, hw, stride, pstride = _gn_shape(x, gamma, groups, film, pre)
    splits = lib.nhmc_gn_splits(B, Cc, groups, hw)
    ws = torch.empty(B * groups * splits * 2, dtype=torch.float64, device=x.device)
    y = torch.empty_like(x)
    rc = lib.nhmc_gn_act_fwd(_p(x, torch.float32, 'x'), _p(gamma, torch.float32, 'gamma'), _p(beta, torch.float32, 'beta'),
                             _ptr(film), stride, _ptr(pre), pstride, float(eps), int(act), _p(y), _p(ws), splits, B, Cc, groups,
                             hw, _stream())
    _lib.check(rc, 'nhmc_gn_act_fwd')
    return y, ws, splits


def gn_act_bwd(x, dy, gamma, beta, groups, eps, act, film, fwd_ws, splits, pre=None, add=None):
    """Input gradient of gn_act_fwd (parameters, FiLM and pre-bias terms are constants of the path).
    add (optional, x's shape): a second gradient of x, added in the same pass."""
    lib = _lib.load()
    B, Cc, hw, stride, pstride = _gn_shape(x, gamma, groups, film, pre)
    ws = torch.empty(B * groups * splits * 2, dtype=torch.float64, device=x.device)
    dx = torch.empty_like(x)
    if add is not None and (add.shape != x.shape or add.data_ptr() == dx.data_ptr()):
        raise _lib.NhmcError('gn_act_bwd: `add` must have the shape of x')
    rc = lib.nhmc_gn_act_bwd(_p(x, torch.float32, 'x'), _p(dy, torch.float32, 'dy'), _p(gamma, torch.float32, 'gamma'),
                             _p(beta, torch.float32, 'beta'), _ptr(film), stride, _ptr(pre), pstride, float(eps), int(act),
                             _p(fwd_ws, torch.float64), _p(add, torch.float32, 'add') if add is not None else _ptr(None), _p(dx),
                             _p(ws), splits, B, Cc, groups, hw, _stream())
    _lib.check(rc, 'nhmc_gn_act_bwd')
    return dx


def bias_add2(h, bias, other):
    """(h + bias[c]) + other for [B, C, ...] tensors: a convolution's bias folded into the residual add after it."""
    lib = _lib.load()
    if h.shape != other.shape or bias.numel() != h.shape[1]:
        raise _lib.NhmcError('bias_add2: shape mismatch')
    out = torch.empty_like(h)
    rc = lib.nhmc_bias_add2(_p(h, torch.float32, 'h'), _p(bias, torch.float32, 'bias'), _p(other, torch.float32, 'other'),
                            _p(out), h.shape[0], h.shape[1], h[0, 0].numel(), _stream())
    _lib.check(rc, 'nhmc_bias_add2')
    return out


def psnr(xt, x_orig):
    lib = _lib.load()
    B, N = _chains_elems(xt)
    ws = torch.empty(B * lib.nhmc_data_tiles(N), dtype=torch.float64, device=xt.device)
    out = torch.empty(B, dtype=torch.float32, device=xt.device)
    _lib.check(lib.nhmc_psnr(_p(xt, torch.float32, 'xt'), _p(x_orig, torch.float32, 'x_orig'), _p(out), _p(ws),
                             B, N, _stream()), 'nhmc_psnr')
    return out


# ---- a1 -------------------------------------------------------------------------------------
def randn_philox(shape, seed, chain_id0, draw, scale=1.0, device='cuda', out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=device)
    B, N = _chains_elems(out)
    _lib.check(lib.nhmc_randn_philox(_p(out, torch.float32), int(seed), int(chain_id0), int(draw), float(scale),
                                     B, N, _stream()), 'nhmc_randn_philox')
    return out


def uniform_philox(n_chains, seed, chain_id0, draw, device='cuda'):
    lib = _lib.load()
    out = torch.empty(n_chains, dtype=torch.float32, device=device)
    _lib.check(lib.nhmc_uniform_philox(_p(out), int(seed), int(chain_id0), int(draw), n_chains, _stream()),
               'nhmc_uniform_philox')
    return out


def copy_probe(src, dst):
    """Streaming copy with the fused update's access pattern (measurement aid for bench.py's copy ceiling)."""
    lib = _lib.load()
    if src.numel() != dst.numel():
        raise _lib.NhmcError('copy_probe: size mismatch')
    _lib.check(lib.nhmc_copy_probe(_p(src, torch.float32, 'src'), _p(dst, torch.float32, 'dst'), src.numel(), _stream()),
               'nhmc_copy_probe')
    return dst
