"""ctypes binding of libnhmc.so (the C ABI declared in include/nhmc.h).

There is no fallback: if the shared object is missing or a symbol cannot be resolved this
module raises, and every product entry point that needs a kernel raises with it.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, 'libnhmc.so')

P, I, I64, D, F, U32, U64, SZ = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float, C.c_uint32, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/nhmc.h one to one
SIGNATURES = {
    'nhmc_abi_version': (I, []),
    'nhmc_status_string': (C.c_char_p, [I]),
    'nhmc_last_launch_error': (C.c_char_p, []),
    'nhmc_leapfrog_tiles': (I, [I64]),
    'nhmc_leapfrog_ws_bytes': (SZ, [I, I64]),
    'nhmc_leapfrog_fused': (I, [I, P, P, P, P, P, P, D, I, I64, P, P]),
    'nhmc_leapfrog_first': (I, [P, P, P, P, P, P, P, D, I, I64, P, P]),
    'nhmc_grad_cache_store': (I, [P, P, P, P, P, P, I, I64, I64, I, I64, P]),
    'nhmc_leapfrog_first_cached': (I, [P, P, P, P, P, I64, P, P, D, I, I64, P, P]),
    'nhmc_leapfrog_last_cached': (I, [P, P, P, P, P, P, I64, P, P, I64, P, P, D, I, I64, P, P]),
    'nhmc_grad_cache_flip': (I, [P, P, I, P]),
    'nhmc_hamiltonian_cached': (I, [P, I, P, P, I64, P, D, P, P, I, P]),
    'nhmc_ddim_mix_fwd': (I, [P, P, I, P, P, I, P, P, P, I, I, I64, P]),
    'nhmc_ddim_map_back': (I, [P, P, P, P, I, I64, P]),
    'nhmc_ddim_mix_bwd': (I, [P, P, P, P, P, I, P, P, I, P, P, I, I, I, I64, P]),
    'nhmc_ddim_mix_bwd_inpaint': (I, [P, P, I, P, P, P, P, I64, P, P, I, P, I, I, I64, P]),
    'nhmc_inpaint_px_tiles': (I, [I, I64]),
    'nhmc_ddim_mix_bwd_inpaint_px': (I, [P, P, I, P, P, P, P, P, I64, P, P, I, P, I, I, I64, P]),
    'nhmc_sr_vjp_tiles': (I, [I, I, I]),
    'nhmc_ddim_mix_bwd_sr': (I, [P, P, I, P, P, P, I, P, P, P, I, I, I, P]),
    'nhmc_data_tiles': (I, [I64]),
    'nhmc_sr_tiles': (I, [I, I, I]),
    'nhmc_data_ws_bytes': (SZ, [I, I64]),
    'nhmc_data_inpaint': (I, [P, P, P, I, P, P, I, I64, I64, P]),
    'nhmc_data_sr': (I, [P, P, I, I, P, P, I, I, I, P]),
    'nhmc_sum_partials': (I, [P, I, I, I, I, P, P]),
    'nhmc_inpaint_H': (I, [P, P, P, I, I64, I64, P]),
    'nhmc_inpaint_Ht': (I, [P, P, P, I, I64, I64, P]),
    'nhmc_sr_H': (I, [P, P, I, I, I, I, P]),
    'nhmc_sr_Ht': (I, [P, P, I, F, I, I, I, P]),
    'nhmc_color_tiles': (I, [I64]),
    'nhmc_data_color': (I, [P, P, P, I, P, P, I, I, I64, P]),
    'nhmc_ddim_mix_bwd_color': (I, [P, P, I, P, P, P, P, P, P, P, I, I, I64, P]),
    'nhmc_color_H': (I, [P, P, P, I, I, I64, P]),
    'nhmc_color_Ht': (I, [P, P, I, P, I, I, I64, P]),
    'nhmc_cs_tiles': (I, [I, I]),
    'nhmc_cs_H': (I, [P, P, P, P, I, I, I, I64, P]),
    'nhmc_cs_Ht': (I, [P, P, P, P, I, I, I, I64, P]),
    'nhmc_data_cs': (I, [P, P, I, P, P, P, I, I, I, P]),
    'nhmc_data_cs_vjp': (I, [P, P, P, P, I, P, P, P, P, P, P, I, I, I, P]),
    'nhmc_spectral_apply': (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    'nhmc_spectral_tiles': (I, [I, I]),
    'nhmc_data_spectral': (I, [P, P, P, P, P, I, P, P, P, I, I, I, P]),
    'nhmc_data_spectral_vjp': (I, [P, P, P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, P]),
    'nhmc_spectral_project': (I, [P, P, P, P, P, I, I, I, P]),
    'nhmc_data_spectral_proj': (I, [P, P, P, P, I, P, P, P, I, I, I, P]),
    'nhmc_data_spectral_proj_vjp': (I, [P, P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, P]),
    'nhmc_sandwich_rect': (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    'nhmc_srconv_tiles': (I, [I, I]),
    'nhmc_data_srconv': (I, [P, P, P, P, P, P, P, I, P, P, P, I, I, I, I, P]),
    'nhmc_data_srconv_vjp': (I, [P, P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, P]),
    'nhmc_hamiltonian': (I, [P, I, P, P, D, P, P, I, P]),
    'nhmc_metropolis': (I, [P, P, P, P, P, P, I, P]),
    'nhmc_schedule_begin': (I, [P, P, P, P, P, P, D, I, I, I, P]),
    'nhmc_accept_commit': (I, [P, P, P, P, P, P, I, I, I, I64, P]),
    'nhmc_schedule_end': (I, [P, P, P, P, P, P, P, P, I, P]),
    'nhmc_latent_commit': (I, [P, P, P, I, I, P, P, P, P, P, I, I64, P]),
    'nhmc_schedule_end_latent': (I, [P, P, P, P, P, P, P, P, D, I, I, P]),
    'nhmc_leapfrog_mass': (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I64, P, P]),
    'nhmc_mass_sort_ws_bytes': (SZ, [I, I64]),
    'nhmc_mass_from_variance': (I, [P, I, P, P, P, P, P, P, SZ, I, I64, P]),
    'nhmc_schedule_begin_mass': (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    'nhmc_vq_nearest': (I, [P, P, P, P, I, I, I64, I, P]),
    'nhmc_gn_splits': (I, [I, I, I, I64]),
    'nhmc_gn_act_fwd': (I, [P, P, P, P, I64, P, I64, F, I, P, P, I, I, I, I, I64, P]),
    'nhmc_gn_act_bwd': (I, [P, P, P, P, P, I64, P, I64, F, I, P, P, P, P, I, I, I, I, I64, P]),
    'nhmc_bias_add2': (I, [P, P, P, P, I, I, I64, P]),
    'nhmc_psnr': (I, [P, P, P, P, I, I64, P]),
    'nhmc_randn_philox': (I, [P, U64, U32, U32, F, I, I64, P]),
    'nhmc_copy_probe': (I, [P, P, I64, P]),
    'nhmc_uniform_philox': (I, [P, U64, U32, U32, I, P]),
}

_lib = None
ABI_VERSION = 2          # NHMC_ABI_VERSION of include/nhmc.h this binding was written against


class NhmcError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises NhmcError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NhmcError(
            f'{LIB_PATH} is missing: the HIP extension has not been built. '
            'Run `python noise-space-hmc_amd/build.py` (or __graft_entry__.build()). '
            'There is no CPU fallback for the sampler kernels.')
    # torch ships its own libamdhip64 (same soname as /opt/rocm's).  Whichever copy is loaded first serves
    # the whole process, and libnhmc must launch on the runtime that owns torch's device context and
    # streams -- so make sure torch's is in before dlopen resolves our NEEDED entry.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise NhmcError(f'{LIB_PATH} does not export {name}; rebuild it') from exc
        fn.restype, fn.argtypes = res, args
    if lib.nhmc_abi_version() != ABI_VERSION:
        raise NhmcError('libnhmc ABI version mismatch; rebuild it')
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().nhmc_status_string(status).decode()
        if status == 4:
            msg += ': ' + load().nhmc_last_launch_error().decode()
        raise NhmcError(f'{what}: {msg} (status {status})')
