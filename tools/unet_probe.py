"""Score-network tuning probe: time U-Net forward + input-gradient at the bench chunk under MIOpen solver-selection
modes and memory formats.  Usage: python tools/unet_probe.py [chunk] > gpurun_out/unet_probe.log
A heartbeat thread keeps gpurun_out/unet_probe.hb fresh while MIOpen searches."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nhmc  # noqa: E402
from nhmc import unet  # noqa: E402


def heartbeat():
    os.makedirs('gpurun_out', exist_ok=True)
    while True:
        with open('gpurun_out/unet_probe.hb', 'w') as f:
            f.write(str(time.time()))
        time.sleep(20)


def run(model, x, t, reps):
    def once():
        xi = x.detach().requires_grad_(True)
        out = model(xi, t)
        (g,) = torch.autograd.grad(out, xi, torch.ones_like(out))
        return g
    t0 = time.time()
    once()
    torch.cuda.synchronize()
    first = time.time() - t0
    once()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    return first, (time.time() - t0) / reps


def main():
    threading.Thread(target=heartbeat, daemon=True).start()
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['default', 'benchmark', 'channels_last', 'both']
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = unet.create_model(**unet.FFHQ_CONFIG).to(dev).eval().requires_grad_(False)
    x = torch.randn(chunk, 3, 256, 256, device=dev)
    t = torch.full((chunk,), 500.0, device=dev)
    for mode in modes:
        torch.backends.cudnn.benchmark = mode in ('benchmark', 'both')
        cl = mode in ('channels_last', 'both')
        m = model.to(memory_format=torch.channels_last) if cl else model.to(memory_format=torch.contiguous_format)
        xi = x.contiguous(memory_format=torch.channels_last) if cl else x
        first, avg = run(m, xi, t, 3)
        print(f'{mode:14s} chunk {chunk}: first {first:.1f} s, fwd+bwd {avg * 1e3:.1f} ms '
              f'({chunk * 2 * 388e9 * 1.0 / avg / 1e12:.0f} TF/s nominal at 2x388 GF/sample)', flush=True)


if __name__ == '__main__':
    main()
