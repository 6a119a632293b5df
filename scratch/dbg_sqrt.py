import sys, torch
sys.path.insert(0, '.')
import nhmc.kernels as K
from oracle import schedule
b = schedule.betas_fp32()
tab = schedule.alpha_bar_table(b)            # 1001 values in (0,1]
B = tab.numel()
ones = torch.ones(B, 1, 2, 2, device='cuda'); zeros = torch.zeros_like(ones)
c3 = K.ddim_map_back(ones, zeros, tab)[:, 0, 0, 0].cpu()
ref3 = tab.sqrt()
print('c3 mismatches', int((c3 != ref3).sum()), 'of', B)
out = K.ddim_mix_fwd(zeros, ones, tab, tab, want=('add_up',))['add_up'][:, 0, 0, 0].cpu()
ref4 = (1 - tab).sqrt()
print('c4 mismatches', int((out != ref4).sum()))
i = torch.nonzero(c3 != ref3).reshape(-1)[:5]
for k in i.tolist():
    print(k, float(tab[k]), c3[k].view(torch.int32).item() - ref3[k].view(torch.int32).item())
# division check: u = (xt - 0)/c2 with e = 0
x = torch.randn(B, 1, 2, 2)
o = K.ddim_mix_fwd(x.cuda(), zeros, tab, tab, want=('x0_t',))['x0_t'].cpu()
ref = (x / tab.sqrt().view(-1, 1, 1, 1)).clip(-1, 1)
print('div mismatches', int((o != ref).sum()), 'of', o.numel())
# torch-on-GPU sqrt for comparison
print('torch gpu sqrt mismatches', int((tab.cuda().sqrt().cpu() != ref3).sum()))
