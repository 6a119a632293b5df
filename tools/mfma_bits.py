"""Is a product of the MFMA GEMM kernels (v_mfma_f32_32x32x2_f32, accumulator from zero, k ascending) the SAME BITS as
torch's CPU matmul?  torch's CPU sgemm is a sequential FMA chain over k (checked against an emulated chain in the
build container), and the MI355X guide lists the f32-input MFMA as bitwise equal to an fmaf chain -- if both hold, a
chain of MFMA products in the reference's stage order reproduces the reference's matmuls exactly.
    python tools/mfma_bits.py"""
import sys
import torch
sys.path.insert(0, '.')
import nhmc.kernels as K



def chain_product(S1, x):
    """S1^T x as an explicit k-ascending FMA chain from a zero accumulator, emulated in float64 on the GPU: a * b is
    exact in float64 and fl32(a * b + acc) is the fused result (double rounding is possible in principle; none was
    seen against torch's CPU sgemm in the build container, where the two agree bit for bit up to K = 256)."""
    a, b = S1.double().cuda(), x.double().cuda()
    acc = torch.zeros(x.shape[0], S1.shape[1], x.shape[2], dtype=torch.float32, device='cuda')
    for k in range(S1.shape[0]):
        acc = (a[k][None, :, None] * b[:, k][:, None, :] + acc.double()).float()
    return acc.cpu()


torch.manual_seed(0)
for (n_img, K1, R1, C1) in ((6, 64, 64, 32), (6, 32, 32, 32), (3, 256, 256, 256), (6, 256, 256, 64), (6, 64, 256, 64)):
    x = torch.randn(n_img, K1, R1)
    S1 = torch.randn(K1, C1)
    eye = torch.eye(R1)
    want = torch.matmul(S1.t(), x)                                    # [n, C1, R1], torch CPU
    got = K.sandwich_rect(x.cuda(), S1.cuda(), eye.cuda()).cpu()      # (x^T S1)^T I = S1^T x
    emu = chain_product(S1, x)
    print(f'   MFMA product == emulated k-ascending FMA chain: {torch.equal(got, emu)} (mismatching {float((got != emu).float().mean()):.4f}); '
          f'this host\'s torch CPU matmul == that chain: {torch.equal(want, emu)}')
    same = torch.equal(got, want)
    print(f'one product  K={K1} R={R1} C={C1}: bit-identical to torch CPU matmul: {same}; mismatching entries '
          f'{float((got != want).float().mean()):.4f}, max rel {float((got - want).abs().max() / want.abs().max()):.2e}')
    S2 = torch.randn(R1, C1)
    want2 = torch.matmul(torch.matmul(S1.t(), x), S2)                 # left first, then right
    got2 = K.sandwich_rect(x.cuda(), S1.cuda(), S2.cuda()).cpu()
    print(f'   sandwich (left product first): bit-identical {torch.equal(got2, want2)}; mismatching {float((got2 != want2).float().mean()):.4f}')
if True:
    from nhmc import operators
    op = operators.build_operator('deblur_aniso', 3, 256, 'cuda')
    x = torch.randn(2, 3, 256, 256)
    f = [m.cpu() for m in op.factors]
    U1, U2, V1, V2 = f[0], f[1], f[2], f[3]
    D = op.Dmap.cpu()
    want = torch.matmul(torch.matmul(U1, D * torch.matmul(torch.matmul(V1.t(), x), V2)), U2.t())
    got = op.H(x.cuda()).cpu().reshape(2, 3, 256, 256)
    print(f'aniso H at 256 (pair kernel): bit-identical to the staged torch CPU matmuls: {torch.equal(got, want)}; mismatching '
          f'{float((got != want).float().mean()):.4f}, max rel {float((got - want).abs().max() / want.abs().max()):.2e}')
