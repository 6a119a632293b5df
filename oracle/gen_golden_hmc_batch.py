"""Generate tests/golden/g17_hmc_f64_inpaint_32_s{1001,1002,1003}.npz: three more reference `hmc()` runs of the G14
inpainting problem (same x, y_0, mask; oracle/gen_golden.py g4_hmc(f64=True)) under other seeds of the momentum / uniform
stream.  The reference is batch-1 only (main_sampling.py:719 raises for n > 1); the build runs B chains at once with
per-chain schedules, so the batched engine is checked by replaying these runs TOGETHER with G14's as four chains of one
call.  Only what differs from G14 is stored (seed, uniforms, -dH, the returned images, first / last momentum).
Build container only; same rules as oracle/gen_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle.gen_golden as gg  # noqa: E402

if __name__ == '__main__':
    ms = gg.import_reference()
    torch.set_num_threads(4)
    base = np.load(os.path.join(gg.OUT, 'g14_hmc_f64_inpaint_32.npz'))
    real_save = gg.save
    for seed in (1001, 1002, 1003):
        def save(name, **arrays):
            for k in ('x', 'y_0', 'x_orig', 'missing'):
                assert np.array_equal(arrays[k], base[k]), k              # the same problem as G14
            keep = {k: arrays[k] for k in ('seed', 'u', 'neg_dH', 'out', 'p0', 'p_last', 'psnr')}
            real_save(name, **keep)
        gg.save = save
        gg.g4_hmc(ms, 'inpaint', 32, seed=seed, f64=True, out_name=f'g17_hmc_f64_inpaint_32_s{seed}.npz')
