"""Which ATen ops are left between the convolutions of the FFHQ U-Net (forward + input gradient) once the fused glue runs:
torch.profiler, grouped by op and input shapes.  Usage: python tools/unet_ops_profile.py [chains]"""
import sys
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from nhmc import unet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
torch.manual_seed(0)
net = unet.create_model(**unet.FFHQ_CONFIG).to(dev).eval().requires_grad_(False)
x = torch.randn(B, 3, 256, 256, device=dev)
t = torch.full((B,), 500.0, device=dev)


def once():
    xi = x.detach().requires_grad_(True)
    out = net(xi, t)
    return torch.autograd.grad(out, xi, torch.ones_like(out))[0]


for _ in range(2):
    once()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    once()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True)
        if e.key.startswith('aten::') and not any(k in e.key for k in ('conv', 'miopen', 'cudnn'))]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f'ATen ops outside the convolutions, {B} chains, one forward + input gradient: {tot / 1e3:.1f} ms of device time')
for e in rows[:28]:
    print(f'{e.self_device_time_total / 1e3:8.2f} ms  x{e.count:<4d} {e.key:28s} {str(e.input_shapes)[:110]}')
