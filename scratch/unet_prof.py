import sys, torch
sys.path.insert(0, '.')
import nhmc.unet as U
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = U.create_model(**U.FFHQ_CONFIG).cuda().eval().requires_grad_(False)
x = torch.randn(B, 3, 256, 256, device='cuda'); t = torch.full((B,), 500.0, device='cuda')
for _ in range(2):
    leaf = x.clone().requires_grad_(True); e = m(leaf, t); (g,) = torch.autograd.grad(e, leaf, torch.ones_like(e))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    leaf = x.clone().requires_grad_(True); e = m(leaf, t); (g,) = torch.autograd.grad(e, leaf, torch.ones_like(e))
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    if 'conv' in ev.key.lower() and ev.device_time_total > 0:
        rows.append((ev.device_time_total, ev.count, ev.key, str(ev.input_shapes)[:150]))
rows.sort(reverse=True)
for r in rows[:40]:
    print(f'{r[0]/1e3:9.2f} ms  x{r[1]:3d}  {r[2]:45s} {r[3]}')
print(prof.key_averages().table(sort_by='device_time_total', row_limit=12, max_name_column_width=70))
