import sys, time, os, torch, torch.nn.functional as F
mode = sys.argv[1]
if mode == 'bench': torch.backends.cudnn.benchmark = True
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
def run(cin, cout, hw, k=3, fmt=torch.contiguous_format, tag=''):
    x = torch.randn(B, cin, hw, hw, device='cuda').to(memory_format=fmt).requires_grad_(True)
    w = torch.randn(cout, cin, k, k, device='cuda').to(memory_format=fmt)
    t0 = time.time()
    y = F.conv2d(x, w, padding=k // 2); torch.cuda.synchronize(); first = time.time() - t0
    go = torch.ones_like(y)
    t0 = time.time(); (g,) = torch.autograd.grad(y, x, go); torch.cuda.synchronize(); firstb = time.time() - t0
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.time(); y = F.conv2d(x, w, padding=k // 2); torch.cuda.synchronize(); t1 = time.time()
        (g,) = torch.autograd.grad(y, x, go); torch.cuda.synchronize(); ts.append((t1 - t0, time.time() - t1))
    f, b = min(t[0] for t in ts), min(t[1] for t in ts)
    fl = 2 * B * cin * cout * k * k * hw * hw
    print(f'{mode}{tag} B={B} {cin}->{cout} k{k} @{hw}: first {first:.2f}s/{firstb:.2f}s fwd {f*1e3:.2f} ms ({fl/f/1e12:.0f} TF/s) bwd-data {b*1e3:.2f} ms ({fl/b/1e12:.0f} TF/s)', flush=True)
for (ci, co, hw, k) in [(128, 128, 256, 3), (256, 128, 256, 3), (3, 128, 256, 3), (128, 6, 256, 3), (128, 128, 128, 3), (256, 256, 64, 3), (256, 128, 256, 1)]:
    run(ci, co, hw, k)
if mode == 'nhwc':
    for (ci, co, hw, k) in [(128, 128, 256, 3), (256, 128, 256, 3)]:
        run(ci, co, hw, k, torch.channels_last, ' nhwc')
