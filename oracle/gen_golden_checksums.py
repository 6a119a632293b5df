"""Generate tests/golden/g18_trace_{deg}_256.npz: a per-call CHECKSUM TRACE of the reference's whole `hmc()` run of G16
(oracle/gen_golden_hmc_256.py: 256 x 256, float64 tiny score, same seeds), so that a replay on the GPU can be compared
with the reference at every score-network call instead of only at the accept decisions and the returned images --
the instrument that locates WHERE a replay leaves the reference's run (tests/test_reference_run_gpu.py, G16 cs4).

What is recorded (everything through `hmc()`'s ARGUMENTS; the reference code is unchanged):
  * the score model is wrapped: for each call the exact integer checksum (sum of the fp32 bit patterns as int64) of its
    input (the noise-space position at DDIM step 0, the intermediate DDIM states at steps 1, 2), of its output, of the
    gradient autograd hands to its output (d loss / d e) and of the gradient that reaches its input;
  * the operator is wrapped: checksum of the image handed to `H` (the clipped decode) and of `H`'s output.
The reference makes 1 + L ladders of 3 score calls per outer iteration (main_sampling.py:693, :709).
Build container only; about 25 minutes on 8 cores.

    python oracle/gen_golden_checksums.py cs4
"""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import import_reference, tiny_model, save, OUT  # noqa: E402
from oracle.tiny_score import F64Score  # noqa: E402


def checksum(t):
    """order-independent, exact: the fp32 bit patterns summed as int64"""
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())


class RecordingScore(torch.nn.Module):
    def __init__(self, net, rec):
        super().__init__()
        self.net, self.rec = net, rec

    def forward(self, x, t):
        i = len(self.rec['score_in'])
        self.rec['score_in'].append(checksum(x))
        self.rec['g_in'].append(0)
        self.rec['g_out'].append(0)
        out = self.net(x, t)
        self.rec['score_out'].append(checksum(out[:, :3]))
        if x.requires_grad:
            x.register_hook(lambda g, i=i: self.rec['g_in'].__setitem__(i, checksum(g)))
        if out.requires_grad:
            out.register_hook(lambda g, i=i: self.rec['g_out'].__setitem__(i, checksum(g[:, :3])))
        return out


class RecordingOperator:
    """forwards everything to the reference operator; records what goes through H"""

    def __init__(self, op, rec):
        self._op, self._rec = op, rec

    def H(self, x):
        self._rec['H_in'].append(checksum(x))
        out = self._op.H(x)
        self._rec['H_out'].append(checksum(out))
        return out

    def __getattr__(self, k):
        return getattr(self._op, k)


def main():
    deg = sys.argv[1] if len(sys.argv) > 1 else 'cs4'
    max_iters = int(sys.argv[2]) if len(sys.argv) > 2 else None
    ms = import_reference()
    from algos.unconditional import Unconditional
    from obs_functions.Hfuncs import WalshHadamardCS, Colorization
    dim, seed = 256, 5678
    if deg == 'cs4':
        perm = torch.randperm(dim * dim, generator=torch.Generator().manual_seed(1600))
        Hf = WalshHadamardCS(3, dim, 4, perm, 'cpu')
    elif deg == 'color':
        Hf = Colorization(dim, 'cpu')
    else:
        raise SystemExit('cs4 | color')
    torch.set_num_threads(8)
    g16 = np.load(os.path.join(OUT, f'g16_hmc_f64_{deg}_256.npz'))
    rec = dict(score_in=[], score_out=[], g_in=[], g_out=[], H_in=[], H_out=[])
    net = RecordingScore(F64Score(tiny_model()), rec)
    b = torch.from_numpy(ms.get_beta_schedule(beta_schedule='linear', beta_start=1e-4, beta_end=0.02,
                                              num_diffusion_timesteps=1000)).float()
    g = torch.Generator().manual_seed(11)                          # oracle/gen_golden.py g4_hmc, same order
    x_orig = torch.rand(1, 3, dim, dim, generator=g) * 2 - 1
    sigma_0 = 2 * 0.05
    y_0 = Hf.H(x_orig).detach()
    y_0 = y_0 + sigma_0 * torch.randn(y_0.shape, generator=g)
    x = torch.randn(1, 3, dim, dim, generator=g)
    assert np.array_equal(y_0.numpy(), g16['y_0']), 'not the G16 problem'
    opt = argparse.Namespace(tau=1.0, epsilon=0.05, m=1.0, sigma_0=sigma_0, algo='hmc', noise='ddpm',
                             image_folder='/tmp/nhmc_golden_scratch')
    os.makedirs(opt.image_folder, exist_ok=True)
    Hrec = RecordingOperator(Hf, rec)
    algo = Unconditional(net, Hrec, sigma_0)
    neg_dH = []
    real_exp = torch.exp

    class Stop(Exception):
        pass

    def exp(t, *a, **k):
        if t.numel() == 1 and t.dim() == 1:
            neg_dH.append(float(t.reshape(-1)[0]))
            if max_iters and len(neg_dH) >= max_iters:
                raise Stop
        return real_exp(t, *a, **k)

    torch.manual_seed(seed)
    torch.exp = exp
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ms.hmc(x, 1, b, [250, 500, 750], [-1, 250, 500], algo, opt, y_0, Hrec, x_orig)
    except Stop:
        pass
    finally:
        torch.exp = real_exp
    n = len(neg_dH)
    assert np.array_equal(np.array(neg_dH), g16['neg_dH'][:n]), 'the instrumented run is not the G16 run'
    arrays = {k: np.array(v, dtype=np.int64) for k, v in rec.items()}
    arrays.update(neg_dH=np.array(neg_dH), ladders_per_iteration=np.array(21), note=np.array(
        'per score call: score_in, score_out (first 3 channels), g_out (d loss/d e, first 3 channels), g_in (gradient reaching the '
        'input); per H call: H_in (clipped decode), H_out.  checksum = sum of fp32 bit patterns as int64'))
    save(f'g18_trace_{deg}_256.npz' if not max_iters else f'g18_trace_{deg}_256_first{max_iters}.npz', **arrays)


if __name__ == '__main__':
    main()
