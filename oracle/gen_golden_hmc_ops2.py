"""Generate tests/golden/g15_hmc_f64_{color,gauss,cs4,box,sr16}_32.npz and g15_hmc_f64_bicubic2_64.npz: the reference's
WHOLE `hmc()` run (main_sampling.py:660-774) with the remaining operators of SURVEY 8 f.3 -- Colorization, Deblurring
(deblur_gauss), WalshHadamardCS, box inpainting, SuperResolution x16, SRConv (bicubic) -- and the tiny score evaluated in
float64, recorded exactly as G14 is (oracle/gen_golden.py g4_hmc(f64=True)).  Build container only; same rules as
oracle/gen_golden.py: the reference is imported from /root/reference, only arrays are written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.gen_golden import g4_hmc, import_reference, np32  # noqa: E402
from oracle.gen_golden_srconv import bicubic  # noqa: E402


def spectral_export(db, dim):
    hw = dim * dim
    sing = db.singulars()
    D = torch.zeros(3, hw)
    for c in range(3):
        D[c, db._perm] = sing[3 * torch.arange(hw) + c]
    return dict(gauss_U=np32(db.U_small), gauss_V=np32(db.V_small), gauss_D=np32(D.reshape(3, dim, dim)))


def main(which):
    ms = import_reference()
    torch.set_num_threads(4)
    from obs_functions.Hfuncs import Colorization, Deblurring, Inpainting, SRConv, SuperResolution, WalshHadamardCS
    dim = 32
    if 'color' in which:
        g4_hmc(ms, 'color', dim, f64=True, op=Colorization(dim, 'cpu'), out_name='g15_hmc_f64_color_32.npz')
    if 'gauss' in which:
        sigma = 10
        pdf = lambda v: torch.exp(torch.Tensor([-0.5 * (v / sigma) ** 2]))
        kernel = torch.Tensor([pdf(-2), pdf(-1), pdf(0), pdf(1), pdf(2)])
        kernel = kernel / kernel.sum()                                  # main_sampling.py:308-314
        db = Deblurring(kernel, 3, dim, 'cpu')
        g4_hmc(ms, 'gauss', dim, f64=True, op=db, out_name='g15_hmc_f64_gauss_32.npz',
               extra=dict(kernel=np32(kernel), **spectral_export(db, dim)))
    if 'cs4' in which:
        perm = torch.randperm(dim * dim, generator=torch.Generator().manual_seed(1500))
        g4_hmc(ms, 'cs4', dim, f64=True, op=WalshHadamardCS(3, dim, 4, perm, 'cpu'), out_name='g15_hmc_f64_cs4_32.npz',
               extra=dict(perm=np32(perm), ratio=np.array(4)))
    if 'box' in which:
        missing = torch.zeros(dim, dim, 3)                              # main_sampling.py:290-299 at 1/8 scale
        missing[6:22, 9:25, :] = 1.0
        idx = torch.nonzero(missing.view(-1)).squeeze(1)
        g4_hmc(ms, 'box', dim, f64=True, op=Inpainting(3, dim, idx, 'cpu'), out_name='g15_hmc_f64_box_32.npz',
               extra=dict(box_missing=np32(idx)))
    if 'sr16' in which:
        g4_hmc(ms, 'sr16', dim, f64=True, op=SuperResolution(3, dim, 16, 'cpu'), out_name='g15_hmc_f64_sr16_32.npz')
    if 'bicubic2' in which:
        kernel = bicubic(2)
        op = SRConv(kernel, 3, 64, 'cpu', stride=2)
        g4_hmc(ms, 'bicubic2', 64, f64=True, op=op, out_name='g15_hmc_f64_bicubic2_64.npz',
               extra=dict(kernel=np32(kernel), factor=np.array(2), srconv_U=np32(op.U_small), srconv_s=np32(op.singulars_small),
                          srconv_V=np32(op.V_small)))


if __name__ == '__main__':
    main(sys.argv[1:] or ['color', 'gauss', 'cs4', 'box', 'sr16', 'bicubic2'])
