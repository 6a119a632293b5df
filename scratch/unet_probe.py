import sys, time, torch
sys.path.insert(0, '.')
import nhmc.unet as U
torch.manual_seed(0)
m = U.create_model(**U.FFHQ_CONFIG).cuda().eval().requires_grad_(False)
fmt = sys.argv[1] if len(sys.argv) > 1 else 'nchw'
if fmt == 'nhwc':
    m = m.to(memory_format=torch.channels_last)
for B in (1, 4, 8, 16):
    x = torch.randn(B, 3, 256, 256, device='cuda')
    t = torch.full((B,), 500.0, device='cuda')
    for rep in range(3):
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); t0 = time.time()
        leaf = x.clone().requires_grad_(True)
        e = m(leaf, t)
        torch.cuda.synchronize(); t1 = time.time()
        (g,) = torch.autograd.grad(e, leaf, torch.ones_like(e))
        torch.cuda.synchronize(); t2 = time.time()
    print(f'{fmt} B={B} fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB', flush=True)
