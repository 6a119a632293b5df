import sys, time, torch
sys.path.insert(0, '.')
import nhmc.kernels as K
from nhmc import operators
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = K.randn_philox((B, 3, 256, 256), 1, 0, 0)
T = B * 3 * 256 * 256 * 4
op = operators.build_operator('deblur_aniso', 3, 256, dev)
y = K.randn_philox((B, 3, 256, 256), 1, 0, 1).reshape(B, -1)
ms = timeit(lambda: op.data_term(x, y, True), 10)
nprod = 4 if op.projected else 8
fl = nprod * B * 3 * 2 * 256 ** 3
print(f'aniso data term B={B}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  ({fl/1e9:.1f} GFLOP executed, {nprod} products)')
ms = timeit(lambda: op.H(x), 10)
print(f'aniso H B={B}: {ms:.3f} ms  {4 * B * 3 * 2 * 256 ** 3/ms/1e9:.1f} TFLOP/s')
op4 = operators.build_operator('sr4', 3, 256, dev)
y4 = torch.randn(B, op4.M, device=dev)
ms = timeit(lambda: op4.data_term(x, y4, True))
print(f'sr4 data term B={B}: {ms*1e3:.1f} us  {(2*T)/ms/1e6:.0f} GB/s (2T)')
op16 = operators.build_operator('sr16', 3, 256, dev)
y16 = torch.randn(B, op16.M, device=dev)
ms = timeit(lambda: op16.data_term(x, y16, True))
print(f'sr16 data term B={B}: {ms*1e3:.1f} us  {(2*T)/ms/1e6:.0f} GB/s (2T)')
opi = operators.build_operator('inpaint_random', 3, 256, dev)
yi = torch.randn(B, opi.M, device=dev)
ms = timeit(lambda: opi.data_term(x, yi, True))
print(f'inpaint data term B={B}: {ms*1e3:.1f} us  {(2*T)/ms/1e6:.0f} GB/s (2T dense)')
for deg, nbytes in (('cs4', None), ('sr_bicubic4', None), ('color', None), ('deblur_gauss', None)):
    o = operators.build_operator(deg, 3, 256, dev)
    yy = torch.randn(B, o.M, device=dev)
    ms = timeit(lambda: o.data_term(x, yy, True), 10)
    print(f'{deg} data term B={B}: {ms*1e3:.1f} us  ({T/ms/1e6:.0f} GB/s per T moved)')
# fused (data term + last DDIM-step VJP) kernels: R xt, R e[:C], W g_xt, W g_e[:C] = 4T
e6 = K.randn_philox((B, 6, 256, 256), 1, 0, 2)
ge = torch.zeros_like(e6)
at = torch.full((B,), 0.5214230418, device=dev)
an = torch.ones(B, device=dev)
for name, o, yy in (('inpaint', opi, yi), ('sr4', op4, y4), ('sr16', op16, y16)):
    ms = timeit(lambda: o.fused_last_vjp(x, e6, at, an, yy, g_e_out=ge))
    print(f'{name} fused last VJP B={B}: {ms*1e3:.1f} us  {(4*T)/ms/1e6:.0f} GB/s (4T)')
cur = K.ddim_mix_fwd(x, e6, at, an, final_clip=True)['xt_next']
def two_kernel():
    l, g = op.data_term(cur, y, apply_clip=False)
    return K.ddim_mix_bwd(g, x, e6, at, an, final_clip=True, g_e_out=ge)
ms2 = timeit(two_kernel, 10)
ms1 = timeit(lambda: op.fused_last_vjp(x, e6, at, an, y, g_e_out=ge, xt_next=cur), 10)
print(f'aniso data term + last VJP B={B}: two kernels {ms2*1e3:.1f} us, fused epilogue {ms1*1e3:.1f} us')
opp = operators.build_operator('deblur_aniso', 3, 256, dev, spectral_projected=True)
ms = timeit(lambda: opp.data_term(x, y, True), 10)
msv = timeit(lambda: opp.fused_last_vjp(x, e6, at, an, y, g_e_out=ge, xt_next=cur), 10)
flp = 4 * B * 3 * 2 * 256 ** 3
print(f'aniso projected (4 products) B={B}: data term {ms*1e3:.1f} us {flp/ms/1e9:.1f} TFLOP/s, + last VJP {msv*1e3:.1f} us {flp/msv/1e9:.1f} TFLOP/s')
