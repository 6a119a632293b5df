"""Attention of the latent path's VQ-f4 decoder (single head over 64 x 64 = 4096 positions, 512 channels, fp32,
forward + input gradient) and of the LDM U-Net (32-channel heads at 32 x 32 / 16 x 16 / 8 x 8): explicit bmm + softmax
(what nhmc.ldm / nhmc.unet issue, as the reference does) against torch's fused scaled_dot_product_attention.
Usage: python tools/attn_probe.py [chains]"""
import sys
import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')


def timeit(f, n=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def explicit(q, k, v, grad):
    w = torch.softmax(torch.bmm(q.transpose(1, 2), k) * (q.shape[1] ** -0.5), dim=2)     # [b, L, L]
    out = torch.bmm(v, w.transpose(1, 2))
    if grad:
        torch.autograd.grad(out, (q, k, v), torch.ones_like(out))
    return out


def fused(q, k, v, grad):
    out = F.scaled_dot_product_attention(q.transpose(1, 2).unsqueeze(1), k.transpose(1, 2).unsqueeze(1), v.transpose(1, 2).unsqueeze(1))
    if grad:
        torch.autograd.grad(out, (q, k, v), torch.ones_like(out))
    return out.squeeze(1).transpose(1, 2)


for name, b, c, L, grad in (('VQ decoder mid attention', B, 512, 4096, True), ('LDM U-Net 32x32 heads', B * 14, 32, 1024, False),
                            ('LDM U-Net 16x16 heads', B * 21, 32, 256, False), ('LDM U-Net 8x8 heads', B * 28, 32, 64, False)):
    q, k, v = (torch.randn(b, c, L, device=dev, requires_grad=grad) for _ in range(3))
    a, f = explicit(q, k, v, False), fused(q, k, v, False)
    err = float((a - f).abs().max() / a.abs().max())
    for label, fn in (('bmm + softmax', explicit), ('scaled_dot_product_attention', fused)):
        try:
            ms = timeit(lambda: fn(q, k, v, grad))
            print(f'{name} [{b},{c},{L}] {"fwd+bwd" if grad else "fwd"}: {label} {ms:.2f} ms', flush=True)
        except Exception as exc:                                   # noqa: BLE001
            print(f'{name}: {label} failed: {type(exc).__name__}: {str(exc)[:120]}', flush=True)
    print(f'   outputs differ by {err:.1e} (relative to max)', flush=True)
