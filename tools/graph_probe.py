"""Eager vs hipGraph replay of one decode+gradient chunk (FFHQ U-Net in the loop), B = 1, 4, 16."""
import sys, time, torch
sys.path.insert(0, '.')
import nhmc.kernels as K
from nhmc import operators, plugin, sampler, schedule, unet
dev = torch.device('cuda')
torch.manual_seed(0)
op = operators.build_operator('inpaint_random', 3, 256, dev, generator=torch.Generator().manual_seed(1))
model = unet.create_model(**unet.FFHQ_CONFIG).to(dev).eval().requires_grad_(False)
algo = plugin.HMC(model, op, 0.1)
b = torch.from_numpy(schedule.get_beta_schedule('linear', beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)).float().to(dev)
seq, seq_next = schedule.timestep_ladder(1000, 3)
for B in (1, 4, 16):
    eng = sampler.LeapfrogEngine(algo.score, op, b, seq, seq_next, dev, chunk=B)
    x = K.randn_philox((B, 3, 256, 256), 1, 0, 0)
    y = torch.randn(B, op.M, device=dev)
    res = {}
    for mode in (False, True):
        for _ in range(2): out = eng.decode_and_grad(x, y, graph=mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = eng.decode_and_grad(x, y, graph=mode)
        torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 5, out)
    same = all(torch.equal(a, b_) for a, b_ in zip(res[False][1], res[True][1]))
    print(f'B={B}: eager {res[False][0]*1e3:.1f} ms  graph {res[True][0]*1e3:.1f} ms  speed-up {res[False][0]/res[True][0]:.2f}x  identical {same}', flush=True)
