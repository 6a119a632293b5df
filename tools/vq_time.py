import sys, torch
sys.path.insert(0, '.')
import nhmc.kernels as K
dev = torch.device('cuda')
for B, D in ((16, 3), (16, 4), (1, 3), (64, 3)):
    z = K.randn_philox((B, D, 64, 64), 3, 0, 0).clamp_(-1, 1) if D == 3 else torch.rand(B, D, 64, 64, device=dev) * 2 - 1
    cb = torch.rand(8192, D, device=dev) * 2 - 1
    for _ in range(3):
        K.vq_nearest(z, cb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        K.vq_nearest(z, cb)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f'B={B} D={D}: {us:.1f} us  {B*4096*8192/us/1e6:.2f} Tpairs/s')
