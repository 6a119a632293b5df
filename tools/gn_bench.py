"""Fused GroupNorm (+ FiLM + SiLU) kernels of the score network against the HBM roofline, at the FFHQ U-Net's shapes
(64 chains).  Forward = stats (R x) + apply (R x, W y) = 3 passes; backward = stats (R x, dy) + apply (R x, dy, W dx) = 5.
Usage: python tools/gn_bench.py [chains]"""
import sys
import torch
sys.path.insert(0, '.')
import nhmc.kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda')


def timeit(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for Cc, res in ((128, 256), (256, 256), (128, 128), (256, 128), (384, 128), (256, 64), (512, 64), (512, 32), (1024, 16)):
    x = torch.randn(B, Cc, res, res, device=dev)
    dy = torch.randn_like(x)
    gamma, beta = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    film = torch.randn(B, 2 * Cc, device=dev)
    nbytes = x.numel() * 4
    tf = timeit(lambda: K.gn_act_fwd(x, gamma, beta, 32, 1e-5, True, film=film))
    y, ws, splits = K.gn_act_fwd(x, gamma, beta, 32, 1e-5, True, film=film)
    tb = timeit(lambda: K.gn_act_bwd(x, dy, gamma, beta, 32, 1e-5, True, film, ws, splits))
    print(f'[{B},{Cc},{res},{res}] {nbytes / 2 ** 30:.2f} GiB  splits {splits}: fwd {tf * 1e3:.0f} us = {3 * nbytes / tf / 1e9:.2f} TB/s (3 passes), '
          f'bwd {tb * 1e3:.0f} us = {5 * nbytes / tb / 1e9:.2f} TB/s (5 passes)', flush=True)
    del x, dy, y
