"""A/B of the fused GroupNorm kernels inside the FFHQ U-Net: forward + input gradient at chunk 32, NHMC_FUSED_GN=1 vs 0."""
import os
import sys
import time
import torch
sys.path.insert(0, '.')
from nhmc import unet

dev = torch.device('cuda')
torch.manual_seed(0)
net = unet.create_model(**unet.FFHQ_CONFIG).to(dev).eval().requires_grad_(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(n, 3, 256, 256, device=dev)
t = torch.full((n,), 500.0, device=dev)
gout = torch.randn(n, 6, 256, 256, device=dev)
res = {}
for mode in ('1', '0', '1', '0'):
    os.environ['NHMC_FUSED_GN'] = mode

    def step():
        xl = x.clone().requires_grad_(True)
        out = net(xl, t)
        (g,) = torch.autograd.grad(out, xl, gout)
        return out, g
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out, g = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res.setdefault(mode, []).append(ms)
    res['out' + mode], res['g' + mode] = out, g
    print(f'NHMC_FUSED_GN={mode}: forward + input gradient of {n} chains: {ms:.1f} ms, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    torch.cuda.reset_peak_memory_stats()
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
print('fused vs ATen: out rel', rel(res['out1'], res['out0']), 'grad rel', rel(res['g1'], res['g0']))
