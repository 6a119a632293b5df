"""Engine overhead probe: one leapfrog step three ways at B chains of 256x256x3 (inpaint_random):
  kernels_only   the HIP kernels of a step called directly on resident buffers (no engine, no autograd)
  engine_resident `LeapfrogEngine.step` with a resident score (bench.py's hot_path_only): what the sampler launches
  engine_trivial `LeapfrogEngine.step` with a two-op differentiable score (adds 4 elementwise torch kernels)
Prints one JSON line.    python tools/engine_overhead.py [B] [chunk]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nhmc.kernels as K  # noqa: E402
from nhmc import sampler  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device('cuda', 0)
prob = bench.build_problem(dev, B, 0, model=torch.nn.Identity())
op = prob['op']
CH, DIM = bench.CH, bench.DIM
eps = torch.full((B,), 0.05, dtype=torch.float64, device=dev)
sig = torch.full((B,), 1.7, dtype=torch.float64, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# ---- kernels only ------------------------------------------------------------------------------------
x, p, y = prob['x'].clone(), prob['p'].clone(), prob['y']
e = K.randn_philox((B, 2 * CH, DIM, DIM), 7, 0, 0, device=dev)
gs = K.randn_philox((B, CH, DIM, DIM), 7, 0, 1, scale=1e-3, device=dev)
ge = [torch.zeros_like(e) for _ in range(3)]
at = [torch.tensor([a], device=dev).expand(B).contiguous() for a in (0.0033001585, 0.0777966604, 0.5214230418)]
atn = [torch.tensor([a], device=dev).expand(B).contiguous() for a in (0.0777966604, 0.5214230418, 1.0)]


def kernels_only():
    cur, ins = x, []
    for s in range(3):
        ins.append(cur)
        cur = K.ddim_mix_fwd(cur, e, at[s], atn[s], final_clip=(s == 2))['xt_next']
    g2 = None
    for s in (2, 1, 0):
        if s == 2:
            _, g, _ = op.fused_last_vjp(ins[s], e, at[s], atn[s], y, g_e_out=ge[s])
        else:
            g, _ = K.ddim_mix_bwd(g, ins[s], e, at[s], atn[s], gout2=g2, g_e_out=ge[s])
        g2 = gs
    K.leapfrog_fused(K.LF_MID, x, p, g, eps, sig, 1.0, g2=g2)


class Cheap(torch.nn.Module):                      # 2 elementwise kernels forward, 2 backward
    def forward(self, xx, t):
        return torch.cat([xx * 0.5, xx * 0.25], dim=1)


eng = sampler.LeapfrogEngine(Cheap(), op, prob['b'], prob['seq'], prob['seq_next'], dev, chunk=chunk)
x2, p2 = prob['x'].clone(), prob['p'].clone()
ws = K.leapfrog_ws(B, x2[0].numel(), dev)
out = dict(B=B, chunk=chunk,
           kernels_only_ms=round(timed(kernels_only), 4),
           engine_resident_ms=round(bench.hot_path_only(dev, prob, B, 50, chunk=chunk)['ms_per_step'], 4),
           engine_trivial_ms=round(timed(lambda: eng.step(K.LF_MID, x2, x2, p2, y, eps, sig, 1.0, ws)), 4))
out['engine_over_kernels'] = round(out['engine_resident_ms'] / out['kernels_only_ms'], 4)
print(json.dumps(out))
