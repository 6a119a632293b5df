"""Build a MIOpen user find-db for the score network's convolutions at the bench batch (no gfx950 find-db ships with
ROCm 7.2: /opt/rocm/share/miopen/db has them for gfx942 and older only, so MIOpen's immediate mode -- what PyTorch uses
with cudnn.benchmark = False -- chooses solvers by heuristics alone).
    python tools/miopen_find.py find  [batch]     Find for every convolution of one forward + input-gradient pass; the db
                                                  (gpurun_out/miopen_db, seeded from noise-space-hmc_amd/miopen_db) is
                                                  written as each Find completes, so a run cut by its time limit keeps
                                                  what it found and the next run continues from there
    python tools/miopen_find.py time  [batch]     forward + input-gradient time with whatever MIOPEN_USER_DB_PATH holds
A heartbeat thread reports the db's size while MIOpen compiles and times candidates."""
import glob
import os
import shutil
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else 'time'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
DB = os.path.join(ROOT, 'gpurun_out', 'miopen_db')
if mode == 'find':
    os.makedirs(DB, exist_ok=True)
    for f in glob.glob(os.path.join(ROOT, 'noise-space-hmc_amd', 'miopen_db', '*')):
        if not os.path.exists(os.path.join(DB, os.path.basename(f))):
            shutil.copy(f, DB)
    os.environ['MIOPEN_USER_DB_PATH'] = DB
    os.environ.setdefault('MIOPEN_FIND_MODE', 'NORMAL')           # a full Find on a db miss (the default hybrid mode skips solvers it would have to compile)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
import nhmc  # noqa: E402,F401
from nhmc import unet  # noqa: E402


def db_lines():
    n = 0
    for f in glob.glob(os.path.join(os.environ.get('MIOPEN_USER_DB_PATH', DB), '*.ufdb.txt')):
        with open(f) as fh:
            n += sum(1 for _ in fh)
    return n


def heartbeat(t0):
    while True:
        print(f'[{time.time() - t0:6.0f} s] find-db entries: {db_lines()}', flush=True)
        time.sleep(30)


def once(model, x, t):
    xi = x.detach().requires_grad_(True)
    out = model(xi, t)
    (g,) = torch.autograd.grad(out, xi, torch.ones_like(out))
    return g


def main():
    t0 = time.time()
    threading.Thread(target=heartbeat, args=(t0,), daemon=True).start()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = unet.create_model(**unet.FFHQ_CONFIG).to(dev).eval().requires_grad_(False)
    x = torch.randn(batch, 3, 256, 256, device=dev)
    t = torch.full((batch,), 500.0, device=dev)
    torch.backends.cudnn.benchmark = mode == 'find'
    once(model, x, t)
    torch.cuda.synchronize()
    print(f'first pass ({mode}, batch {batch}): {time.time() - t0:.1f} s, MIOPEN_USER_DB_PATH={os.environ.get("MIOPEN_USER_DB_PATH")}, '
          f'entries {db_lines()}', flush=True)
    once(model, x, t)
    torch.cuda.synchronize()
    t1 = time.time()
    for _ in range(3):
        once(model, x, t)
    torch.cuda.synchronize()
    print(f'forward + input gradient, batch {batch}: {(time.time() - t1) / 3 * 1e3:.1f} ms', flush=True)


if __name__ == '__main__':
    main()
