"""Probe: seconds of sampler.hmc_chains per run at the configured step size (rejected) and at 1/1000 of it (accepted)."""
import sys, time, types, math
sys.path.insert(0, '.')
import torch
import bench
from nhmc import sampler
dev = torch.device('cuda')
B = 64
prob = bench.build_problem(dev, B, 0)
x_orig = prob['x'].clamp(-1, 1)
L = 5
for name, tau_, eps_ in (('configured', 0.25, bench.EPS), ('small', (L + 0.5) * bench.EPS * 1e-3, bench.EPS * 1e-3),
                         ('configured', 0.25, bench.EPS), ('small', (L + 0.5) * bench.EPS * 1e-3, bench.EPS * 1e-3)):
    opt = types.SimpleNamespace(tau=tau_, epsilon=eps_, m=1.0, sigma_0=2 * prob['sigma0'])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = sampler.hmc_chains(prob['x'], prob['b'], prob['seq'], prob['seq_next'], prob['algo'], opt, prob['y'], prob['op'], x_orig,
                             noise=sampler.PhiloxNoise(5678, 0), chunk=None, max_iters=2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{name}: {dt:.2f} s, ladders {res.ladders}, ms/ladder {1e3*dt/res.ladders:.1f}, accepted {int(res.n_accept.sum())}, '
          f'peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB', flush=True)
