"""Noise-space HMC sampler: `hmc(...)` with the reference signature, on the HIP kernels.

Replaces main_sampling.py:660-774 (`hmc`) and :898-915 (`iterative_sampling`).  What changes:

  * B independent chains per call (the reference raises for n > 1, main_sampling.py:719): per-chain
    loss, Hamiltonian, accept test, epoch counter, sigma_y / eps / tau and reject counter, all chains
    advancing one trajectory per iteration.  At n = 1 it takes the reference's decisions.
  * One trajectory is sync-free: schedule, momentum draw, L x (decode + gradient + fused update),
    both Hamiltonians, Metropolis test, accept commit and sample collection are kernels on the
    stream; the host reads back one small status vector per trajectory (termination test / logging).
  * The decode's autograd is unrolled by hand: the score network is the only autograd graph; the DDIM
    mix, the final clip, the operator and the residual are closed-form kernels forward and backward,
    and the two gradient contributions reaching a DDIM input (direct + through the score network) are
    summed inside the next kernel instead of by a separate add.

The score network itself stays a PyTorch-ROCm module (north_star): `algo.model(xt, t)`.
"""
import math
import os
from types import SimpleNamespace

import torch

from . import kernels as K
from .plugin import HMC
from .schedule import compute_alpha, mass_tables as _mass_tables


# --------------------------------------------------------------------------------------------- #
# reference-surface decode (kept for plugins that only implement cal_x0 / map_back)
# --------------------------------------------------------------------------------------------- #
def iterative_sampling(xt, n, b, seq, seq_next, algo, opt, y_0, tqdm_disable=True):
    """main_sampling.py:898-915, through the plugin surface."""
    for i, j in zip(reversed(seq), reversed(seq_next)):
        t = (torch.ones(n) * i).to(xt.device)
        next_t = (torch.ones(n) * j).to(xt.device)
        at = compute_alpha(b, t.long())
        at_next = compute_alpha(b, next_t.long())
        x0_t, add_up = algo.cal_x0(xt, t, at, at_next, y_0, getattr(opt, 'noise', 'ddpm'))
        xt = algo.map_back(x0_t, y_0, add_up, at_next, at)
    return xt


# --------------------------------------------------------------------------------------------- #
# noise sources
# --------------------------------------------------------------------------------------------- #
class TorchNoise:
    """The reference's draws: device randn for the momentum (main_sampling.py:692), CPU rand for the
    accept uniform (:720) -- one uniform per chain."""

    def momentum(self, it, like, scale):
        return torch.randn_like(like) * scale

    def uniform(self, it, n, device):
        return torch.rand(n).to(device)


class PhiloxNoise:
    """Counter-based Philox4x32-10 keyed by (seed, global chain id, draw): a chain's noise does not
    depend on how chains are split over launches or ranks."""

    def __init__(self, seed, chain_id0=0):
        self.seed, self.chain_id0 = int(seed), int(chain_id0)

    def momentum(self, it, like, scale):
        return K.randn_philox(like.shape, self.seed, self.chain_id0, it, scale=scale, device=like.device)

    def uniform(self, it, n, device):
        return K.uniform_philox(n, self.seed, self.chain_id0, it, device=device)


class TapeNoise:
    """Momenta / uniforms supplied by the caller (parity tests): p(it) -> [B,C,H,W], u(it) -> [B]."""

    def __init__(self, p, u):
        self.p, self.u = p, u

    def momentum(self, it, like, scale):
        return (self.p(it).to(like.device) * scale).contiguous()

    def uniform(self, it, n, device):
        return self.u(it).to(device=device, dtype=torch.float32).contiguous()


# --------------------------------------------------------------------------------------------- #
# engine
# --------------------------------------------------------------------------------------------- #
class LeapfrogEngine:
    """Decode + data term + hand-unrolled backward for one ladder of DDIM steps.

    score      callable (xt, t) -> [B, C or 2C, H, W], differentiable w.r.t. xt (a torch module)
    operator   object with data_term(xt, y, apply_clip) -> (loss[B] fp64, g_xt)   (nhmc.operators)
    b          betas, fp32 [T] on the device
    chunk      chains per score-network call (activation memory of 3 graphs scales with it)
    """

    def __init__(self, score, operator, b, seq, seq_next, device, chunk=None, alpha_table=None, image_map=None):
        """alpha_table (optional, instead of betas `b`): entry k = alpha-bar at t = k-1, as the latent driver builds
        it from the model's buffers (main_sampling_latent.py:771-773).  image_map (optional): differentiable torch
        map applied to the clipped decode before the operator (the latent variant's first-stage decoder, :651)."""
        self.score, self.operator, self.device, self.chunk = score, operator, device, chunk
        self.image_map = image_map
        self.fuse_last = True                  # use operator.fused_last_vjp when it exists (inpainting)
        self._graphs = {}                      # (n, shape of y) -> captured decode+gradient of one chunk
        self._ge = {}                          # score-output shape -> persistent g_e buffers (sigma-channels stay zero)
        self._outs = None                      # batch-wide decode / loss buffers of `step`
        self.update_events = None              # bench: list collecting (start, end, chains) event pairs of each update launch
        self.n_ladders = 0                     # decode + gradient ladders launched (one per chunk) ...
        self.n_chain_ladders = 0               # ... and the chains they carried (score evaluations = n_steps x this)
        steps = list(zip(reversed(seq), reversed(seq_next)))
        from .schedule import alpha_bar_table
        table = alpha_table.to(device).float() if alpha_table is not None else alpha_bar_table(b)
        idx = torch.tensor([[i + 1, j + 1] for i, j in steps], device=table.device)
        self.t_values = [float(i) for i, _ in steps]
        self.at = [table[idx[s, 0]].reshape(1).to(device) for s in range(len(steps))]
        self.at_next = [table[idx[s, 1]].reshape(1).to(device) for s in range(len(steps))]
        self.n_steps = len(steps)
        self._per_n = {}                       # chunk size -> per-step (t, at, at_next) device arrays of that length

    def _tables(self, n):
        """Per-chain timestep / alpha-bar arrays for a chunk of n chains, built once per n (no per-step fill kernels)."""
        hit = self._per_n.get(n)
        if hit is None:
            hit = self._per_n[n] = [(torch.full((n,), self.t_values[s], device=self.device),
                                     self.at[s].expand(n).contiguous(), self.at_next[s].expand(n).contiguous())
                                    for s in range(self.n_steps)]
        return hit

    def _chunks(self, B):
        c = self.chunk or B
        return [(s, min(B, s + c)) for s in range(0, B, c)]

    def _out_buffers(self, x):
        """Batch-wide decode / loss buffers, kept per shape: the kernels of every chunk write their slice directly."""
        key = (tuple(x.shape), x.device)
        if self._outs is None or self._outs[0] != key:
            self._outs = (key, torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float64, device=x.device))
        return self._outs[1], self._outs[2]

    def decode_and_grad(self, x, y, graph=False):
        """-> xt [B,C,H,W] (clipped decode), loss [B] fp64, (g_direct, g_score): the two pieces of
        d(sum_b loss_b)/dx, summed later inside the consuming kernel (g_score is None for a score evaluated
        without gradient).  Batch-wide tensors: with more than one chunk the per-chunk gradients are gathered by a
        copy -- the sampler itself uses `step`, which hands each chunk's gradients straight to the fused update.

        graph=True replays one hipGraph per chunk (score network forward + input-gradient included) instead of
        launching its ~10^3 kernels from Python: same kernels, same results; pays off when a chunk is small
        (the reference's one-chain operating point), where the step is launch-bound."""
        B = x.shape[0]
        xt_out = torch.empty_like(x)
        loss = torch.empty(B, dtype=torch.float64, device=x.device)
        chunks = self._chunks(B)
        run = self._graphed_chunk if graph else self._decode_and_grad_chunk
        if len(chunks) == 1 and not graph:
            ga, gb = run(x, y, xt_out, loss)
            return xt_out, loss, ga, gb
        ga, gb = torch.empty_like(x), None
        for lo, hi in chunks:
            a, b_ = run(x[lo:hi], y[lo:hi], xt_out[lo:hi], loss[lo:hi])
            ga[lo:hi].copy_(a)
            if b_ is not None:
                gb = torch.empty_like(x) if gb is None else gb
                gb[lo:hi].copy_(b_)
        return xt_out, loss, ga, gb

    def prime(self, cache, x, y, graph=False):
        """Decode + gradient at x into the cache slot `sel` of every chain: the once-per-run evaluation at the start
        point (every later trajectory finds its first gradient in the cache)."""
        xt_out, loss = self._out_buffers(x)
        run = self._graphed_chunk if graph else self._decode_and_grad_chunk
        for lo, hi in self._chunks(x.shape[0]):
            ga, gb = run(x[lo:hi], y[lo:hi], xt_out[lo:hi], loss[lo:hi])
            K.grad_cache_store(ga, gb, loss[lo:hi], cache.g[:, lo:hi], cache.loss[:, lo:hi], cache.sel[lo:hi])
        cache.valid = True

    def step(self, mode, x_in, x, p, y, eps, sigma_y, m_inv, ws, graph=False, cache=None, want_loss=None):
        """Decode + gradient at x_in, then the fused leapfrog update of (x, p), chunk by chunk: every kernel writes
        its slice of the batch-wide outputs and each chunk's two gradient pieces go straight into the update --
        no gather copies, no clone of the position.  mode FIRST: out of place (x_in -> x, x_in kept); MID / LAST:
        x_in is x.  cache (a GradCache, LAST only): the end point's loss and summed gradient are also stored in the
        chains' free cache slots.  want_loss (default: every mode but MID): the per-chain loss is summed from the data term's
        tile partials only when somebody reads it -- the Hamiltonians at a trajectory's two ends (main_sampling.py:697,717);
        a MID step needs the gradient alone and skips that launch (its `loss` return is then stale).
        -> (xt, loss): batch-wide buffers owned by the engine, valid until the next step."""
        B, N = x.shape[0], x[0].numel()
        xt_out, loss = self._out_buffers(x)
        want_loss = (mode != K.LF_MID) if want_loss is None else want_loss
        tiles2 = 2 * K.leapfrog_tiles(N)
        run = self._graphed_chunk if graph else self._decode_and_grad_chunk
        for lo, hi in self._chunks(B):
            ga, gb = run(x_in[lo:hi], y[lo:hi], xt_out[lo:hi], loss[lo:hi] if want_loss else K.NO_LOSS)
            w = ws[lo * tiles2: hi * tiles2]
            if self.update_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), hi - lo)
                ev[0].record()
            if mode == K.LF_FIRST:
                K.leapfrog_first(x_in[lo:hi], x[lo:hi], p[lo:hi], ga, eps[lo:hi], sigma_y[lo:hi], m_inv, w, g2=gb)
            elif mode == K.LF_LAST and cache is not None:
                K.leapfrog_last_cached(x[lo:hi], p[lo:hi], ga, gb, cache.g[:, lo:hi], cache.loss[:, lo:hi],
                                       cache.sel[lo:hi], loss[lo:hi], eps[lo:hi], sigma_y[lo:hi], m_inv, w)
            else:
                K.leapfrog_fused(mode, x[lo:hi], p[lo:hi], ga, eps[lo:hi], sigma_y[lo:hi], m_inv, w, g2=gb)
            if self.update_events is not None:
                ev[1].record()
                self.update_events.append(ev)
        return xt_out, loss

    def _graphed_chunk(self, x, y, xt_out, loss_out):
        key = (tuple(x.shape), tuple(y.shape))
        rec = self._graphs.get(key)
        if rec is None:
            sx, sy = x.clone(), y.clone()
            sxt, sloss = torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                       # warm-up: solver selection, allocator pools
                for _ in range(2):
                    self._decode_and_grad_chunk(sx, sy, sxt, sloss)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                grads = self._decode_and_grad_chunk(sx, sy, sxt, sloss)
            rec = self._graphs[key] = (g, sx, sy, sxt, sloss, grads)
        g, sx, sy, sxt, sloss, grads = rec
        sx.copy_(x)
        sy.copy_(y)
        g.replay()
        xt_out.copy_(sxt)
        if loss_out is not K.NO_LOSS:
            loss_out.copy_(sloss)
        return grads                                            # the graph's own buffers: valid until its next replay

    def _decode_and_grad_chunk(self, x, y, xt_out, loss_out):
        """Writes the clipped decode and the per-chain loss of this chunk into xt_out / loss_out (views of the
        batch-wide buffers) and returns the chunk's gradient pieces (g_direct, g_score or None)."""
        n = x.shape[0]
        S = self.n_steps
        self.n_ladders += 1
        self.n_chain_ladders += n
        tab = self._tables(n)
        ins, outs = [], []
        cur = x
        for s in range(S):
            leaf = cur.detach().requires_grad_(True)
            with torch.enable_grad():
                e = self.score(leaf, tab[s][0])
            e_c = e.detach() if e.is_contiguous() else e.detach().contiguous()
            cur = K.ddim_mix_fwd(leaf.detach(), e_c, tab[s][1], tab[s][2],
                                 final_clip=(s == S - 1), out=xt_out if s == S - 1 else None)['xt_next']
            ins.append((leaf, e_c))
            outs.append(e if e.requires_grad else None)           # None: score evaluated without gradient (latent model)
        fused = self.image_map is None and self.fuse_last and hasattr(self.operator, 'fused_last_vjp') \
            and outs[S - 1] is not None
        # the final clip is applied by the last mix; its mask is re-derived inside the last mix backward
        if fused:
            g = None
        elif self.image_map is None:
            _, g = self.operator.data_term(cur, y, apply_clip=False, loss_out=loss_out)
        else:
            zleaf = cur.detach().requires_grad_(True)
            with torch.enable_grad():
                img = self.image_map(zleaf)
            img_c = img.detach()
            _, g_img = self.operator.data_term(img_c if img_c.is_contiguous() else img_c.contiguous(), y,
                                               apply_clip=False, loss_out=loss_out)
            (g,) = torch.autograd.grad(img, zleaf, g_img)
            g = g if g.is_contiguous() else g.contiguous()
        g2 = None
        # one persistent score-gradient buffer per DDIM step and shape: its sigma-channels are zeroed once and never
        # rewritten (the mix VJP writes only the first C channels: -T of traffic per step)
        bufs = None
        if any(o is not None for o in outs):
            key = tuple(ins[0][1].shape)
            bufs = self._ge.get(key)
            if bufs is None:
                bufs = self._ge[key] = [torch.zeros_like(ins[0][1]) for _ in range(S)]
        for s in reversed(range(S)):
            leaf, e_c = ins[s]
            if fused and s == S - 1:                  # data term + last-step VJP in one kernel
                extra = dict(xt_next=cur) if getattr(self.operator, 'fused_wants_decode', False) else {}
                _, g_direct, g_e = self.operator.fused_last_vjp(leaf.detach(), e_c, tab[s][1], tab[s][2], y,
                                                                g_e_out=bufs[s], loss_out=loss_out, **extra)
            else:
                g_direct, g_e = K.ddim_mix_bwd(g, leaf.detach(), e_c, tab[s][1], tab[s][2],
                                               final_clip=(s == S - 1), gout2=g2, want_g_e=outs[s] is not None,
                                               g_e_out=bufs[s] if outs[s] is not None else None)
            if outs[s] is not None:
                (g_score,) = torch.autograd.grad(outs[s], leaf, g_e)
                outs[s] = None
                g2 = g_score if g_score.is_contiguous() else g_score.contiguous()
            else:
                g2 = None
            g = g_direct
        return g, g2

    @torch.no_grad()
    def decode(self, x):
        """Forward only (no gradient): the clipped decode."""
        cur = x
        n, S = x.shape[0], self.n_steps
        for s in range(S):
            t = torch.full((n,), self.t_values[s], device=x.device)
            e = self.score(cur, t).contiguous()
            cur = K.ddim_mix_fwd(cur, e, self.at[s].expand(n), self.at_next[s].expand(n), final_clip=(s == S - 1))['xt_next']
        return cur


class ChainState:
    """Per-chain sampler state, resident on the device (row a7)."""

    FIELDS_I32 = ('epoch', 'rejected', 'active', 'n_accept', 'n_reject')
    FIELDS_F64 = ('tau', 'eps', 'sigma_y', 'eps_eff')

    def __init__(self, n, tau, eps, device):
        self.t = {k: torch.zeros(n, dtype=torch.int32, device=device) for k in self.FIELDS_I32}
        self.t.update({k: torch.zeros(n, dtype=torch.float64, device=device) for k in self.FIELDS_F64})
        self.t['tau'].fill_(tau)
        self.t['eps'].fill_(eps)
        self.t['sigma_y'].fill_(1.0)

    def __getitem__(self, k):
        return self.t[k]

    def get(self, k, default=None):
        return self.t.get(k, default)

    def view(self, n):
        """The first n chains' state (same storage): what the kernels see after compaction."""
        v = object.__new__(ChainState)
        v.t = {k: a[:n] for k, a in self.t.items()}
        return v

    def permute(self, order):
        self.t = {k: a.index_select(0, order) for k, a in self.t.items()}


class GradCache:
    """Loss and summed gradient at every chain's accepted position, and room for the next proposal's: two slots per
    chain, `sel[c]` = the slot that belongs to chain c's accepted position (include/nhmc.h, "Gradient cache").

    The reference evaluates decode + loss + gradient at x at the top of every outer iteration (main_sampling.py:693-695)
    -- a point it already evaluated: in the previous iteration's last leapfrog step if that proposal was accepted
    (:709-711 at x_proposal, then x = x_proposal :731), in the previous iteration's own :693-695 if it was rejected.
    Neither loss nor gradient depends on sigma_y / eps (they enter in the update and in H), so the values are
    bit-identical; with the cache a trajectory costs L score ladders instead of L + 1."""

    def __init__(self, x):
        B = x.shape[0]
        self.g = torch.empty((2,) + tuple(x.shape), dtype=torch.float32, device=x.device)
        self.loss = torch.zeros(2, B, dtype=torch.float64, device=x.device)
        self.sel = torch.zeros(B, dtype=torch.int32, device=x.device)
        self.valid = False

    def view(self, n):
        """The first n chains' share (the active prefix after compaction): same storage."""
        v = object.__new__(GradCache)
        v.g, v.loss, v.sel, v.valid = self.g[:, :n], self.loss[:, :n], self.sel[:n], self.valid
        return v

    def permute(self, order):
        self.g = self.g.index_select(1, order)
        self.loss = self.loss.index_select(1, order)
        self.sel = self.sel.index_select(0, order)


def run_trajectory(engine, x, p, y, state, m, L, ws=None, graph=False, x_prop=None, cache=None):
    """One outer iteration (main_sampling.py:693-718) for all chains.  x is NOT modified; p is updated in place.
    x_prop: optional buffer for the proposal (kept by the caller across trajectories; allocated here otherwise).
    cache: a GradCache kept by the caller across trajectories.  With it the first half step takes loss and gradient
    at x from the cache (primed here on first use) instead of decoding x again, and the last step leaves the end
    point's in the chains' free slots; the CALLER flips `cache.sel` for accepted chains (K.grad_cache_flip).
    -> dict(x_prop, p, xt, loss, H0, H1); xt / loss are the engine's buffers, valid until its next step."""
    B, N = x.shape[0], x[0].numel()
    m_inv = m ** (-1)
    eps, sig = state['eps_eff'], state['sigma_y']
    ws = ws if ws is not None else K.leapfrog_ws(B, N, x.device)
    x_prop = x_prop if x_prop is not None else torch.empty_like(x)
    if cache is None:
        xt, loss = engine.step(K.LF_FIRST, x, x_prop, p, y, eps, sig, m_inv, ws, graph=graph)
        H0 = K.hamiltonian(ws, N, loss, sig, m_inv)
    else:
        if not cache.valid:
            engine.prime(cache, x, y, graph=graph)
        K.leapfrog_first_cached(x, x_prop, p, cache.g, cache.sel, eps, sig, m_inv, ws)
        H0 = K.hamiltonian(ws, N, cache.loss, sig, m_inv, sel=cache.sel)
    for l in range(L):
        xt, loss = engine.step(K.LF_MID if l < L - 1 else K.LF_LAST, x_prop, x_prop, p, y, eps, sig, m_inv, ws,
                               graph=graph, cache=cache)
    H1 = K.hamiltonian(ws, N, loss, sig, m_inv)
    return dict(x_prop=x_prop, p=p, xt=xt, loss=loss, H0=H0, H1=H1)


def hmc_chains(x, b, seq, seq_next, algo, opt, y_0, H_funcs, x_orig=None, *, noise=None, epochs=60, sampling=20,
               chunk=None, max_iters=None, log=None, collect_trace=False, graph=False, reuse=True, compact=True,
               compact_quantum=None):
    """Per-chain HMC for B = x.shape[0] chains (graph=True: hipGraph replay of each decode+gradient chunk).

    reuse    the first half step of a trajectory takes loss and gradient at x from the GradCache instead of decoding
             x again: L score ladders per trajectory (+ one per run) instead of L + 1, same bits.
    compact  chains that reached their last epoch leave the batch: the per-chain state is kept partitioned (running
             chains first, in their original order) and every kernel and score call runs on the running prefix,
             rounded up to a multiple of `compact_quantum` chains (default: the score chunk, else B // 8) so that the
             score network sees few distinct batch sizes.  Noise is still drawn for all B chains in chain order, so a
             chain's run does not depend on when the others finish.
    Returns a SimpleNamespace (per-chain results in the caller's chain order):
        samples [B, sampling, C, H, W]   accepted decodes of epochs epochs+sampling .. epochs+2*sampling-1
        x       [B, C, H, W]             final noise-space positions
        n_accept, n_reject, epoch [B]    int32
        psnr    [B] or None              PSNR of the last accepted decode against x_orig
        iters   int                      trajectories run
        ladders, chain_ladders           decode+gradient ladders launched / chains they carried (score calls = 3 x)
        chain_trajectories               sum over trajectories of the chains that ran it
        trace   list of dicts (dH, accept, epoch per iteration) when collect_trace
    """
    device = x.device
    B, N = x.shape[0], x[0].numel()
    tau, epsilon, m = float(opt.tau), float(opt.epsilon), float(getattr(opt, 'm', 1.0))
    sigma_0 = float(opt.sigma_0)
    L = max(1, math.floor(tau / epsilon))                                   # :664, fixed for the run
    total = epochs + 2 * sampling
    noise = noise or TorchNoise()
    operator = H_funcs if hasattr(H_funcs, 'data_term') else None
    if operator is None:
        raise TypeError('H_funcs must be an nhmc.operators operator (needs the fused data_term); '
                        'wrap other linear operators before calling hmc()')
    score = algo.score if hasattr(algo, 'score') else algo.model
    engine = LeapfrogEngine(score, operator, b, seq, seq_next, device, chunk=chunk)
    x = x.detach().clone().contiguous()
    y_0 = y_0.contiguous()
    x_orig = x_orig.contiguous() if x_orig is not None else None
    state = ChainState(B, tau, epsilon, device)
    samples = torch.zeros((B, sampling) + tuple(x.shape[1:]), dtype=torch.float32, device=device)
    xt_last = torch.zeros_like(x)
    ws = K.leapfrog_ws(B, N, device)
    x_prop = torch.empty_like(x)
    cache = GradCache(x) if reuse else None
    if cache is not None:
        engine.prime(cache, x, y_0, graph=graph)
    quantum = int(compact_quantum or (chunk if chunk and chunk < B else max(1, B // 8)))
    ids, ids_dev, n_run = list(range(B)), None, B       # slot -> chain id; chains [0, n_run) of the slot order run
    chain_trajectories = 0
    trace = [] if collect_trace else None
    it = 0

    def in_chain_order(v, fill=0):
        """[n_run] or [B] slot-ordered device vector -> [B] host vector in the caller's chain order."""
        v = v.detach().cpu()
        if v.numel() < B:
            v = torch.cat([v, torch.full((B - v.numel(),), fill, dtype=v.dtype)])
        if ids_dev is None:
            return v.clone()
        out = torch.empty_like(v)
        out[torch.tensor(ids)] = v
        return out

    while True:
        sv = state if n_run == B else state.view(n_run)
        K.schedule_begin(sv, sigma_0, epochs, sampling)
        p = noise.momentum(it, x, math.sqrt(m))                            # all B chains, chain order
        u = noise.uniform(it, B, device)
        if ids_dev is not None:
            p, u = p.index_select(0, ids_dev[:n_run]), u.index_select(0, ids_dev[:n_run])
        out = run_trajectory(engine, x[:n_run], p, y_0[:n_run], sv, m, L, ws, graph=graph, x_prop=x_prop[:n_run],
                             cache=cache.view(n_run) if cache is not None else None)
        accept, dH = K.metropolis(out['H0'], out['H1'], u, sv['active'])
        K.accept_commit(accept, sv['epoch'], x[:n_run], out['x_prop'], out['xt'], samples[:n_run], epochs, sampling)
        K.accept_commit(accept, sv['epoch'], xt_last[:n_run], out['xt'], None, None, epochs, sampling)
        if cache is not None:
            K.grad_cache_flip(accept, cache.sel[:n_run])
        epoch_before = sv['epoch'].clone() if (collect_trace or log) else None
        K.schedule_end(accept, sv)
        it += 1
        chain_trajectories += n_run
        # the one host read per trajectory: termination, compaction (and optional logging)
        status = torch.stack([sv['epoch'], accept]).cpu()
        if collect_trace:
            trace.append(dict(dH=in_chain_order(dH), accept=in_chain_order(accept), epoch=in_chain_order(epoch_before, total),
                              sigma_y=in_chain_order(state['sigma_y']), eps=in_chain_order(state['eps'])))
        if log is not None:
            log(it, status, sv, out, x_orig[:n_run] if x_orig is not None else None, ids[:n_run])
        ep = status[0].tolist()
        n_act = sum(1 for e in ep if e < total)
        if n_act == 0 or (max_iters is not None and it >= max_iters):
            break
        want = min(n_run, -(-n_act // quantum) * quantum)
        if compact and want < n_run:
            order = [s_ for s_ in range(n_run) if ep[s_] < total] + [s_ for s_ in range(n_run) if ep[s_] >= total] + \
                list(range(n_run, B))
            od = torch.tensor(order, device=device)
            x, y_0, samples, xt_last = (t.index_select(0, od) for t in (x, y_0, samples, xt_last))
            x_orig = x_orig.index_select(0, od) if x_orig is not None else None
            state.permute(od)
            if cache is not None:
                cache.permute(od)
            ids = [ids[s_] for s_ in order]
            ids_dev = torch.tensor(ids, device=device)
            n_run = want
    if ids_dev is not None:                                                 # back to the caller's chain order
        inv = torch.empty(B, dtype=torch.int64)
        inv[torch.tensor(ids)] = torch.arange(B)
        inv = inv.to(device)
        x, samples, xt_last = (t.index_select(0, inv) for t in (x, samples, xt_last))
        x_orig = x_orig.index_select(0, inv) if x_orig is not None else None
        state.permute(inv)
    psnr = K.psnr(xt_last, x_orig) if x_orig is not None else None
    return SimpleNamespace(samples=samples, x=x, n_accept=state['n_accept'], n_reject=state['n_reject'],
                           epoch=state['epoch'], psnr=psnr, iters=it, trace=trace, xt=xt_last, L=L,
                           ladders=engine.n_ladders, chain_ladders=engine.n_chain_ladders,
                           chain_trajectories=chain_trajectories)


def hmc(x, n, b, seq, seq_next, algo, opt, y_0, H_funcs, x_orig):
    """Reference entry point (main_sampling.py:660, called at :483).

    n == 1: returns torch.stack of the 20 collected decodes, [20, C, H, W] -- what the reference returns.
    n  > 1: (the reference raises) returns [n, 20, C, H, W].
    Side effects kept: one stdout line per accepted epoch of chain 0 ('epoch', PSNR, sigma_y, tau) unless
    opt.quiet; the per-epoch PNGs are written only when opt.save_images is set (default off: encoding
    a PNG per accept was a host sync per trajectory in the reference)."""
    quiet = bool(getattr(opt, 'quiet', False))
    noise = getattr(opt, 'noise_source', None)
    if noise is None:
        seed = getattr(opt, 'philox_seed', None)
        noise = PhiloxNoise(seed, getattr(opt, 'chain_id0', 0)) if seed is not None else TorchNoise()

    def log(it, status, state, out, x_orig_, ids):
        if quiet or 0 not in ids:
            return
        s0 = ids.index(0)                                  # chain 0's slot (compaction may have moved it)
        if not bool(status[1][s0]):
            return
        ps = K.psnr(out['xt'][s0:s0 + 1].contiguous(), x_orig_[s0:s0 + 1].contiguous())
        print('epoch', int(status[0][s0]), 'PSNR:', float(ps[0]), 'sigma_y:', float(state['sigma_y'][s0]),
              'tau:', float(state['tau'][s0]))
        if getattr(opt, 'save_images', False):
            _save_png(out['xt'][s0], os.path.join(opt.image_folder, f'hmc_{int(status[0][s0])}.png'))

    every = int(getattr(opt, 'progress_every', 0) or 0)

    def progress(it, status, state, out, x_orig_, ids):
        if every and it % every == 0:
            import sys
            print(f'[hmc] trajectory {it}: {status.shape[1]} chains in the batch, epochs min {int(status[0].min())} '
                  f'max {int(status[0].max())}, accepted this round {int(status[1].sum())}/{status.shape[1]}',
                  file=sys.stderr, flush=True)

    res = hmc_chains(x, b, seq, seq_next, algo, opt, y_0, H_funcs, x_orig, noise=noise,
                     epochs=int(getattr(opt, 'hmc_epochs', 60)), sampling=int(getattr(opt, 'hmc_sampling', 20)),
                     chunk=getattr(opt, 'score_chunk', None), log=(progress if every else None) if quiet else log,
                     graph=bool(getattr(opt, 'use_graph', False)), reuse=bool(getattr(opt, 'reuse_gradient', True)),
                     compact=bool(getattr(opt, 'compact_chains', True)))
    return res.samples[0] if n == 1 else res.samples


# --------------------------------------------------------------------------------------------- #
# diagonal-mass variant
# --------------------------------------------------------------------------------------------- #
def hmc_mass_chains(x, b, seq, seq_next, algo, opt, y_0, H_funcs, x_orig=None, *, noise=None, burn=5, epochs=40,
                    sampling=10, chunk=None, max_iters=None, collect_trace=False, mass_tables=None, reuse=True):
    """Per-chain form of `hmc_test_conditioning` (main_sampling.py:776-894): HMC with a diagonal mass rebuilt from
    the rank transform of each accepted trajectory's position variance.  Returns SimpleNamespace(samples
    [B, 4*sampling-sampling... = total-(epochs+sampling), C, H, W], x, epoch, n_accept, n_reject, iters, trace)."""
    device = x.device
    B, N = x.shape[0], x[0].numel()
    tau, epsilon = float(opt.tau), float(opt.epsilon)
    sigma_0 = float(opt.sigma_0)
    L = max(1, math.floor(tau / epsilon))
    total = burn + epochs + 4 * sampling
    n_keep = total - (epochs + sampling)
    noise = noise or TorchNoise()
    if not hasattr(H_funcs, 'data_term'):
        raise TypeError('H_funcs must be an nhmc.operators operator (needs the fused data_term)')
    score = algo.score if hasattr(algo, 'score') else algo.model
    engine = LeapfrogEngine(score, H_funcs, b, seq, seq_next, device, chunk=chunk)
    sig = [sigma_0 + 0.9 if e < burn else sigma_0 + 0.9 * (1 - (e - burn) / epochs) ** 3 for e in range(epochs)] + [sigma_0]
    sigma_table = torch.tensor(sig, dtype=torch.float64, device=device)                      # :808-813 in Python floats
    x = x.detach().clone().contiguous()
    y_0 = y_0.contiguous()
    st = ChainState(B, tau, epsilon, device)
    st.t['welford_on'] = torch.zeros(B, dtype=torch.int32, device=device)
    inv_m, std_m = torch.ones_like(x), torch.ones_like(x)
    mean, m2 = torch.zeros_like(x), torch.zeros_like(x)
    samples = torch.zeros((B, n_keep) + tuple(x.shape[1:]), dtype=torch.float32, device=device)
    ws, sort_ws = K.leapfrog_ws(B, N, device), None
    # sqrt(M_r), 1 / M_r by rank r: constants of N, built once with the reference's own tensor expressions (:863-868).
    # mass_tables = (std, inv) overrides them (a parity test hands in the tables of the host the reference ran on).
    tables = tuple(t.to(device=device, dtype=torch.float32).contiguous() for t in mass_tables) if mass_tables is not None \
        else _mass_tables(N, device)
    # (loss, summed gradient) at the accepted position, slot 0 of a GradCache whose selector stays 0: :823-825 re-evaluates a
    # point the previous iteration already evaluated (see GradCache); slot 1 takes the end point of every trajectory and
    # is copied over slot 0 for the chains that accept
    cache = GradCache(x) if reuse else None
    trace = [] if collect_trace else None
    it = 0
    while True:
        K.schedule_begin_mass(st, sigma_table, burn, epochs, sampling)
        z = noise.momentum(it, x, 1.0)
        eps, sy, won = st['eps_eff'], st['sigma_y'], st['welford_on']
        if cache is None:
            xt, loss, ga, gb = engine.decode_and_grad(x, y_0)
        else:
            if not cache.valid:
                engine.prime(cache, x, y_0)
            loss, ga, gb = cache.loss[0], cache.g[0], None
        x_prop, p = x.clone(), torch.empty_like(x)
        m2.zero_()                                      # chains without Welford keep M2 = 0 (the reference's zeros, :802)
        K.leapfrog_mass(K.LF_FIRST, x_prop, p, ga, inv_m, eps, sy, ws, g2=gb, z=z, std_m=std_m)
        H0 = K.hamiltonian(ws, N, loss, sy, 1.0)
        for l in range(L):
            xt, loss, ga, gb = engine.decode_and_grad(x_prop, y_0)
            K.leapfrog_mass(K.LF_MID if l < L - 1 else K.LF_LAST, x_prop, p, ga, inv_m, eps, sy, ws, g2=gb,
                            welford_on=won, mean=mean, m2=m2, l=l)
        H1 = K.hamiltonian(ws, N, loss, sy, 1.0)
        u = noise.uniform(it, B, device)
        accept, dH = K.metropolis(H0, H1, u, st['active'])
        if cache is not None:
            K.grad_cache_store(ga, gb, loss, cache.g, cache.loss, cache.sel, flip=1)          # end point -> slot 1
            K.accept_commit(accept, st['epoch'], cache.g[0], cache.g[1], None, None, 0, 0)    # accepted chains: slot 1 -> slot 0
            cache.loss[0].copy_(torch.where(accept.bool(), cache.loss[1], cache.loss[0]))
        # mass rebuild for accepted chains past epochs//3 (pre-increment epoch, :857)
        flags = (accept * (st['epoch'] > epochs // 3).int()).contiguous()
        sort_ws = K.mass_from_variance(m2, L, flags, inv_m, std_m, tables, sort_ws)
        K.accept_commit(accept, st['epoch'], x, x_prop, xt, samples, epochs + sampling - n_keep, n_keep)
        epoch_before = st['epoch'].clone() if collect_trace else None
        K.schedule_end(accept, st)
        it += 1
        status = torch.stack([st['epoch'], accept]).cpu()
        if collect_trace:
            trace.append(dict(dH=dH.cpu(), accept=status[1].clone(), epoch=epoch_before.cpu()))
        if int(status[0].min()) >= total or (max_iters is not None and it >= max_iters):
            break
    return SimpleNamespace(samples=samples, x=x, epoch=st['epoch'], n_accept=st['n_accept'], n_reject=st['n_reject'],
                           iters=it, trace=trace, L=L)


def hmc_test_conditioning(x, n, b, seq, seq_next, algo, opt, y_0, H_funcs, x_orig):
    """Reference entry point (main_sampling.py:776): [35, C, H, W] at n = 1, [n, 35, C, H, W] otherwise."""
    noise = getattr(opt, 'noise_source', None)
    if noise is None:
        seed = getattr(opt, 'philox_seed', None)
        noise = PhiloxNoise(seed, getattr(opt, 'chain_id0', 0)) if seed is not None else TorchNoise()
    res = hmc_mass_chains(x, b, seq, seq_next, algo, opt, y_0, H_funcs, x_orig, noise=noise,
                          chunk=getattr(opt, 'score_chunk', None))
    return res.samples[0] if n == 1 else res.samples


# --------------------------------------------------------------------------------------------- #
# latent variant
# --------------------------------------------------------------------------------------------- #
def hmc_latent_chains(x, seq, seq_next, algo, opt, y_0, H_funcs, x_orig=None, *, noise=None, epochs=50, sampling=10,
                      chunk=None, collect_trace=False, reuse=True):
    """Per-chain form of `hmc_latent` (main_sampling_latent.py:623-762).  The epoch index is the shared loop
    counter (a reject consumes its epoch, :646,733); sigma_y / tau / eps / reject counter / sample ring are per
    chain.  Returns SimpleNamespace(samples [B, <=sampling latents...] as a list per chain, x, n_accept, trace)."""
    device = x.device
    B, N = x.shape[0], x[0].numel()
    model = algo.model
    tau, epsilon, m = float(opt.tau), float(opt.epsilon), float(getattr(opt, 'm', 1.0))
    sigma_0, sigma_y0 = float(opt.sigma_0), float(opt.sigma_y)
    L = max(1, math.floor(tau / epsilon))
    noise = noise or TorchNoise()
    if not hasattr(H_funcs, 'data_term'):
        raise TypeError('H_funcs must be an nhmc.operators operator (needs the fused data_term)')
    table = torch.cat([model.alphas_cumprod_prev[0:1], model.alphas_cumprod], dim=0)          # :771-773
    engine = LeapfrogEngine(algo.score, H_funcs, None, seq, seq_next, device, chunk=chunk, alpha_table=table,
                            image_map=model.differentiable_decode_first_stage)
    x = x.detach().clone().contiguous()
    y_0 = y_0.contiguous()
    st = ChainState(B, tau, epsilon, device)
    st.t['sigma_y'].fill_(sigma_y0)
    st.t['count'] = torch.zeros(B, dtype=torch.int32, device=device)
    st.t['has_prev'] = torch.zeros(B, dtype=torch.int32, device=device)
    ring = torch.zeros((B, sampling) + tuple(x.shape[1:]), dtype=torch.float32, device=device)
    xt_last = torch.zeros_like(x)
    ws = K.leapfrog_ws(B, N, device)
    x_prop = torch.empty_like(x)
    cache = GradCache(x) if reuse else None          # :650-652 re-evaluates a point the previous iteration already has
    trace = [] if collect_trace else None
    for epoch in range(epochs + 2 * sampling):
        st['eps_eff'].copy_(st['eps'])
        p = noise.momentum(epoch, x, math.sqrt(m))
        out = run_trajectory(engine, x, p, y_0, st, m, L, ws, x_prop=x_prop, cache=cache)
        u = noise.uniform(epoch, B, device)
        accept, dH = K.metropolis(out['H0'], out['H1'], u, None)
        if cache is not None:
            K.grad_cache_flip(accept, cache.sel)
        final = epoch >= epochs
        sig_next = sigma_0 if final else sigma_y0 * (sigma_0 / sigma_y0) ** (epoch / epochs)   # :693-695,705-706
        if collect_trace:
            trace.append(dict(dH=dH.cpu(), accept=accept.cpu(), sigma_y=st['sigma_y'].cpu().clone(), eps=st['eps'].cpu().clone()))
        K.latent_commit(accept, st, final, sampling, x, out['x_prop'], xt_last, out['xt'], ring)
        K.schedule_end_latent(accept, st, sig_next, final)
    count = st['count'].cpu().tolist()
    samples = []
    for c in range(B):                                                       # last `sampling` pushes, oldest first
        k = min(count[c], sampling)
        order = [(count[c] - k + j) % sampling for j in range(k)]
        samples.append(ring[c, order])
    return SimpleNamespace(samples=samples, x=x, n_accept=st['n_accept'], count=count, trace=trace, xt=xt_last, L=L,
                           ladders=engine.n_ladders, chain_ladders=engine.n_chain_ladders)


def hmc_latent(x, n, seq, seq_next, algo, opt, y_0, H_funcs, x_orig):
    """Reference entry point (main_sampling_latent.py:623, called at :466): returns the stack of the last 10
    collected latents at n = 1; a list of such stacks for n > 1 (the reference raises there)."""
    noise = getattr(opt, 'noise_source', None)
    if noise is None:
        seed = getattr(opt, 'philox_seed', None)
        noise = PhiloxNoise(seed, getattr(opt, 'chain_id0', 0)) if seed is not None else TorchNoise()
    res = hmc_latent_chains(x, seq, seq_next, algo, opt, y_0, H_funcs, x_orig, noise=noise,
                            chunk=getattr(opt, 'score_chunk', None))
    return res.samples[0] if n == 1 else res.samples


def _save_png(img, path):
    """inverse_data_transform + 8-bit PNG (the reference uses torchvision.utils.save_image)."""
    from PIL import Image
    arr = ((img.detach().clamp(-1, 1) + 1) * 127.5).round().byte().permute(1, 2, 0).cpu().numpy()
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    Image.fromarray(arr).save(path)
