"""Replay a G16 reference run on the GPU against its per-call checksum trace (tests/golden/g18_trace_*.npz,
oracle/gen_golden_checksums.py) and report the FIRST quantity that differs from the reference, with its place in the run
(trajectory, leapfrog step, DDIM step, which tensor).  Every accept decision is given to the reference, so the replay can
only leave the reference's run through a value, never through a decision.

    python tools/trace_replay.py cs4 [--runs 2] [--iters N]

--runs 2 also compares two GPU replays with each other (run-to-run determinism of the GPU path)."""
import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEQ, SEQ_NEXT = [250, 500, 750], [-1, 250, 500]
T = torch.from_numpy


def checksum(t):
    return t.detach().contiguous().view(torch.int32).to(torch.int64).sum()


class RecordingScore(torch.nn.Module):
    def __init__(self, net, rec):
        super().__init__()
        self.net, self.rec = net, rec

    def forward(self, x, t):
        i = len(self.rec['score_in'])
        self.rec['score_in'].append(checksum(x))
        self.rec['g_out'].append(None)
        out = self.net(x, t)
        self.rec['score_out'].append(checksum(out[:, :3]))
        if out.requires_grad:
            out.register_hook(lambda g, i=i: self.rec['g_out'].__setitem__(i, checksum(g[:, :3])))
        return out


def replay(deg, n_iters, golden_dir, return_result=False):
    import nhmc.operators as ops
    from nhmc import plugin, sampler
    from oracle import schedule as osched
    from oracle.tiny_score import F64Score, TinyScore
    g = dict(np.load(os.path.join(golden_dir, f'g16_hmc_f64_{deg}_256.npz'), allow_pickle=False))
    dim, dev = 256, torch.device('cuda')
    if deg == 'cs4':
        op = ops.WalshHadamardCS(3, dim, 4, torch.randperm(dim * dim, generator=torch.Generator().manual_seed(1600)), dev)
    elif deg == 'color':
        op = ops.Colorization(dim, dev)
    else:
        raise SystemExit('cs4 | color')
    y_0 = T(g['y_0'])
    gi = torch.Generator().manual_seed(11)
    x_orig = torch.rand(1, 3, dim, dim, generator=gi) * 2 - 1
    torch.randn(y_0.shape, generator=gi)
    x = torch.randn(1, 3, dim, dim, generator=gi)
    n = min(n_iters or len(g['u']), len(g['u']))
    torch.manual_seed(int(g['seed']))
    P = []
    for _ in range(n):
        P.append(torch.randn(1, 3, dim, dim))
        torch.rand(1)
    prob = np.minimum(1.0, np.exp(np.minimum(g['neg_dH'], 50.0)))
    ref_acc = g['u'] < prob
    u_play = np.where(ref_acc, 0.0, 1.0).astype(np.float32)            # every decision is the reference's
    net = TinyScore()
    net.load_state_dict(torch.load(os.path.join(golden_dir, 'tiny_score.pt'), weights_only=True))
    rec = dict(score_in=[], score_out=[], g_out=[], H_in=[])
    score = RecordingScore(F64Score(net.eval().requires_grad_(False)).to(dev), rec)
    real_vjp = op.fused_last_vjp

    def vjp(*a, **k):
        rec['H_in'].append(checksum(k['xt_next']) if 'xt_next' in k else None)
        return real_vjp(*a, **k)
    op.fused_last_vjp = vjp
    algo = plugin.HMC(score, op, float(g['sigma_0']))
    opt = types.SimpleNamespace(tau=float(g['tau']), epsilon=float(g['epsilon']), m=float(g['m']), sigma_0=float(g['sigma_0']), quiet=True)
    noise = sampler.TapeNoise(lambda it: P[min(it, n - 1)], lambda it: torch.tensor([u_play[min(it, n - 1)]]))
    res = sampler.hmc_chains(x.to(dev), osched.betas_fp32().to(dev), SEQ, SEQ_NEXT, algo, opt, y_0.to(dev), op, x_orig.to(dev),
                             noise=noise, collect_trace=True, max_iters=n)
    out = {k: torch.stack([v if v is not None else torch.zeros((), dtype=torch.int64, device=dev) for v in vals]).cpu().numpy()
           for k, vals in rec.items() if vals and any(v is not None for v in vals)}
    out['dH'] = np.array([float(t['dH'][0]) for t in res.trace])
    out['accept'] = np.array([bool(t['accept'][0]) for t in res.trace])
    op.fused_last_vjp = real_vjp
    return (out, g, n, res) if return_result else (out, g, n)


def first_difference(got, ref, n, L=20):
    """The engine runs 1 ladder for the start point and then L per trajectory (the first half step takes its gradient from
    the cache); the reference runs 1 + L per trajectory, the first of which repeats a point it already evaluated."""
    names = ('score_in', 'score_out', 'g_out')
    events = []                                                        # in execution order
    for it in range(n):
        for j in range(1, L + 1):
            ref_l = it * (L + 1) + j
            got_l = 1 + it * L + (j - 1)
            for s in range(3):
                events.append((it, j, s, 'score_in', got['score_in'][3 * got_l + s], ref['score_in'][3 * ref_l + s]))
                events.append((it, j, s, 'score_out', got['score_out'][3 * got_l + s], ref['score_out'][3 * ref_l + s]))
            if 'H_in' in got:
                events.append((it, j, 3, 'H_in (clipped decode)', got['H_in'][got_l], ref['H_in'][ref_l]))
            # g_out (d loss / d e) is recorded on both sides but not compared: at the last DDIM step its add_up path is
            # sqrt(1 - 1) * g = +-0, and the checksum of bit patterns tells -0.0 from +0.0 (autograd and the fused VJP
            # kernel sum those zeros in different orders; no non-zero value depends on it)
    for k, (it, j, s, what, a, b) in enumerate(events):
        if int(a) != int(b):
            return dict(event=k, trajectory=it, leapfrog_step=j, ddim_step=s, quantity=what, of=len(events))
    return None


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('deg')
    ap.add_argument('--runs', type=int, default=1)
    ap.add_argument('--iters', type=int, default=None)
    a = ap.parse_args()
    gold = os.path.join(ROOT, 'tests', 'golden')
    ref = dict(np.load(os.path.join(gold, f'g18_trace_{a.deg}_256.npz'), allow_pickle=False))
    runs = []
    for r in range(a.runs):
        got, g, n = replay(a.deg, a.iters, gold)
        runs.append(got)
        n = min(n, len(ref['neg_dH']))
        d = first_difference(got, ref, n)
        dev_dH = np.abs(got['dH'][:n] + g['neg_dH'][:n])
        small = np.abs(g['neg_dH'][:n]) < 50
        print(f'run {r}: {n} trajectories replayed; first difference from the reference: {d}', flush=True)
        lim = d['trajectory'] if d else n
        if lim:
            print(f'   max |dH - dH_ref| before it: {np.max(dev_dH[:lim][small[:lim]]):.4f}; after it: '
                  f'{np.max(dev_dH[lim:][small[lim:]]) if lim < n and small[lim:].any() else float("nan"):.4f}', flush=True)
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, 'gpurun_out', f'trace_replay_{a.deg}_run{r}.npz'), **got)
    for r in range(1, len(runs)):
        same = all(np.array_equal(runs[0][k], runs[r][k]) for k in runs[0])
        print(f'GPU run 0 vs run {r}: {"identical in every recorded quantity" if same else "DIFFERENT"}')
        if not same:
            for k in runs[0]:
                if not np.array_equal(runs[0][k], runs[r][k]):
                    print('   first difference in', k, 'at index', int(np.argmax(runs[0][k] != runs[r][k])))
