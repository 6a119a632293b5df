"""A/B of the spectral chain: two products per launch (k_pair256, default) vs one product per launch
(NHMC_SPECTRAL_PAIRS=0).  Run once per setting; the second run compares its outputs with the first run's file.
    NHMC_SPECTRAL_PAIRS=0 python tools/pair_bench.py && python tools/pair_bench.py"""
import os
import sys
import torch
sys.path.insert(0, '.')
import nhmc.kernels as K
from nhmc import operators

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pairs = os.environ.get('NHMC_SPECTRAL_PAIRS', '1') != '0'


def timeit(f, n=20, warm=3):          # NHMC_PAIR_BENCH_LONG=1: 300 warm-up calls, 200 timed (the chip at its loaded clock)
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


x = K.randn_philox((B, 3, 256, 256), 1, 0, 0)
y = K.randn_philox((B, 3, 256, 256), 1, 0, 1).reshape(B, -1)
e6 = K.randn_philox((B, 6, 256, 256), 1, 0, 2)
ge = torch.zeros_like(e6)
at = torch.full((B,), 0.5214230418, device=dev)
an = torch.ones(B, device=dev)
torch.set_num_threads(4)
op = operators.build_operator('deblur_aniso', 3, 256, dev)
cur = K.ddim_mix_fwd(x, e6, at, an, final_clip=True)['xt_next']
fl = 8 * B * 3 * 2 * 256 ** 3
out = {}
out['H'] = op.H(x)
out['loss'], out['g'] = op.data_term(x, y, True)
out['loss2'], out['gx'], _ = op.fused_last_vjp(x, e6, at, an, y, g_e_out=ge, xt_next=cur)
out['ge'] = ge.clone()
ms_h = timeit(lambda: op.H(x), 10)
ms_d = timeit(lambda: op.data_term(x, y, True), 10)
ms_v = timeit(lambda: op.fused_last_vjp(x, e6, at, an, y, g_e_out=ge, xt_next=cur), 10)
if os.environ.get('NHMC_PAIR_BENCH_LONG'):                            # the same after 300 warm-up calls, 200 timed: the chip at its loaded clock
    ms_h = timeit(lambda: op.H(x), 200, 300)
    ms_d = timeit(lambda: op.data_term(x, y, True), 200, 300)
    ms_v = timeit(lambda: op.fused_last_vjp(x, e6, at, an, y, g_e_out=ge, xt_next=cur), 200, 300)
print(f'pairs={int(pairs)} B={B}: H {ms_h*1e3:.1f} us ({fl/2/ms_h/1e9:.1f} TFLOP/s)  data term {ms_d*1e3:.1f} us ({fl/ms_d/1e9:.1f} TFLOP/s)  '
      f'data term + last VJP {ms_v*1e3:.1f} us ({fl/ms_v/1e9:.1f} TFLOP/s)')
path = '/tmp/nhmc_pair_ab.pt'          # hundreds of MB: keep it out of gpurun_out
if os.path.exists(path):
    ref = torch.load(path)
    for k in out:
        same = torch.equal(out[k].cpu(), ref[k])
        rel = float((out[k].cpu().double() - ref[k].double()).abs().max() / ref[k].double().abs().max())
        print(f'  {k}: bit-identical to the other setting: {same} (rel {rel:.2e})')
else:
    torch.save({k: v.cpu() for k, v in out.items()}, path)
