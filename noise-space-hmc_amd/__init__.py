"""nhmc -- MI355X-native noise-space HMC sampler (hot path of Sunsett5/Noise-space-HMC, `--algo hmc`).

Host side in Python (the reference is Python), kernels in HIP for gfx950 behind the C ABI in
include/nhmc.h.  Import as `nhmc` (see /nhmc.py); the directory keeps the project's name.

    kernels    torch-tensor front ends of the C ABI
    schedule   beta / alpha-bar tables and the timestep ladder
    operators  Inpainting / SuperResolution / Deblurring2D with the reference's H / Ht / H_pinv surface
    plugin     Base_Algo surface + the HMC plugin (cal_x0 / map_back on the HIP kernels)
    sampler    hmc(...) with the reference signature, iterative_sampling, the per-chain engine
    unet       guided-diffusion FFHQ U-Net architecture (PyTorch-ROCm; loads ffhq_10m.pt)
    sharding   chain partition over ranks + the single end-of-run gather
"""
__version__ = '0.1.0'

from . import _lib  # noqa: F401


def library_path():
    return _lib.LIB_PATH
