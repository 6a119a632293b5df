"""Can two score-network chunks overlap on the GPU?  64 chains: (a) eager, chunk 32, one stream (the default);
(b) hipGraph replay, chunk 16, one stream; (c) hipGraph replay, chunk 16, two streams (two graph instances)."""
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
B = 64
x = K.randn_philox((B, 3, 256, 256), 1, 0, 0)
y = torch.randn(B, op.M, device=dev)

def timeit(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

e32 = sampler.LeapfrogEngine(algo.score, op, b, seq, seq_next, dev, chunk=32)
print(f'(a) eager chunk 32, 1 stream : {timeit(lambda: e32.decode_and_grad(x, y))*1e3:.0f} ms', flush=True)
del e32; torch.cuda.empty_cache()
e16 = sampler.LeapfrogEngine(algo.score, op, b, seq, seq_next, dev, chunk=16)
print(f'(b) graph chunk 16, 1 stream : {timeit(lambda: e16.decode_and_grad(x, y, graph=True))*1e3:.0f} ms', flush=True)
engs = [e16, sampler.LeapfrogEngine(algo.score, op, b, seq, seq_next, dev, chunk=16)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [torch.empty_like(x), torch.empty(B, dtype=torch.float64, device=dev), torch.empty_like(x), torch.empty_like(x)]

def two_streams():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for i, lo in enumerate(range(0, B, 16)):
        s = streams[i % 2]
        with torch.cuda.stream(s):
            engs[i % 2]._graphed_chunk(x[lo:lo + 16], y[lo:lo + 16], outs[0][lo:lo + 16], outs[1][lo:lo + 16],
                                       outs[2][lo:lo + 16], outs[3][lo:lo + 16])
    for s in streams: cur.wait_stream(s)

print(f'(c) graph chunk 16, 2 streams: {timeit(two_streams)*1e3:.0f} ms', flush=True)
