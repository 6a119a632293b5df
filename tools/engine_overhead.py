"""Engine overhead probe: one leapfrog step (decode + gradient + fused update) through LeapfrogEngine with a trivial
differentiable score, next to bench.py's hot_path_only loop (kernels only).  python tools/engine_overhead.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nhmc.kernels as K  # noqa: E402
from nhmc import sampler  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
prob = bench.build_problem(dev, B, 0)


class Cheap(torch.nn.Module):                      # 2 elementwise kernels forward, 2 backward
    def forward(self, x, t):
        return torch.cat([x * 0.5, x * 0.25], dim=1)


eng = sampler.LeapfrogEngine(Cheap(), prob['op'], prob['b'], prob['seq'], prob['seq_next'], dev, chunk=None)
x, p, y = prob['x'].clone(), prob['p'].clone(), prob['y']
eps = torch.full((B,), 0.05, dtype=torch.float64, device=dev)
sig = torch.full((B,), 1.7, dtype=torch.float64, device=dev)


def step():
    xt, loss, ga, gb = eng.decode_and_grad(x, y)
    K.leapfrog_fused(K.LF_MID, x, p, ga, eps, sig, 1.0, g2=gb)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 50 * 1e3
hot = bench.hot_path_only(dev, prob, B, 50)
print(f'B={B}: engine step with a trivial score {ms:.3f} ms; kernels-only loop {hot["ms_per_step"]:.3f} ms')
