"""CPU: the C-ABI library loads, exports every symbol include/nhmc.h declares, validates its
arguments before touching the device, and the Python binding has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'nhmc.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nhmc_[A-Za-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    import nhmc
    return nhmc._lib.load()


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert len(names) >= 25
    for must in ('nhmc_leapfrog_fused', 'nhmc_ddim_mix_fwd', 'nhmc_ddim_mix_bwd', 'nhmc_data_inpaint', 'nhmc_data_sr',
                 'nhmc_data_spectral', 'nhmc_hamiltonian', 'nhmc_metropolis', 'nhmc_randn_philox'):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    import nhmc
    raw = ctypes.CDLL(nhmc.library_path())
    for name in declared_functions():
        assert hasattr(raw, name), f'{name} declared in include/nhmc.h but not exported by libnhmc.so'
    assert sorted(nhmc._lib.SIGNATURES) == declared_functions(), 'binding table and header disagree'


def test_host_only_entry_points(lib):
    assert lib.nhmc_abi_version() == 2
    assert lib.nhmc_status_string(0) == b'ok' and b'aligned' in lib.nhmc_status_string(2)
    n = 3 * 256 * 256
    assert lib.nhmc_leapfrog_tiles(n) == 96 and lib.nhmc_leapfrog_tiles(3 * 16 * 16) == 1
    assert lib.nhmc_leapfrog_ws_bytes(64, n) == 64 * 96 * 2 * 8
    assert lib.nhmc_data_tiles(n) == 96 and lib.nhmc_sr_tiles(3, 256, 4) == 48 and lib.nhmc_spectral_tiles(3, 256) == 12
    assert lib.nhmc_data_ws_bytes(2, n) >= 2 * 96 * 8


def test_argument_validation_happens_before_any_launch(lib):
    """Null / misaligned / mis-shaped arguments are refused with a status, with no device work."""
    P = ctypes.c_void_p
    null, a16, a4 = P(0), P(0x1000), P(0x1004)
    assert lib.nhmc_leapfrog_fused(1, null, a16, a16, null, a16, a16, 1.0, 1, 1024, null, null) == 1      # ARG
    assert lib.nhmc_leapfrog_fused(7, a16, a16, a16, null, a16, a16, 1.0, 1, 1024, a16, null) == 1        # bad mode
    assert lib.nhmc_leapfrog_fused(0, a16, a16, a16, null, a16, a16, 1.0, 1, 1024, null, null) == 1       # FIRST needs ws
    assert lib.nhmc_leapfrog_fused(1, a4, a16, a16, null, a16, a16, 1.0, 1, 1024, null, null) == 2        # ALIGN
    assert lib.nhmc_leapfrog_fused(1, a16, a16, a16, null, a16, a16, 1.0, 1, 1023, null, null) == 2       # n % 4
    assert lib.nhmc_leapfrog_fused(1, a16, a16, a16, null, a16, a16, 1.0, 70000, 1024, null, null) == 3   # SHAPE
    assert lib.nhmc_ddim_mix_fwd(a16, a16, 5, a16, a16, 0, a16, null, null, 1, 3, 256, null) == 3         # e_channels
    assert lib.nhmc_ddim_mix_fwd(a16, a16, 6, a16, a16, 0, null, null, null, 1, 3, 256, null) == 1        # no output
    assert lib.nhmc_ddim_mix_bwd(a16, a16, a16, a16, a16, 6, a16, a16, 0, a16, a16, 1, 1, 3, 256, null) == 1  # split + gout2
    assert lib.nhmc_data_sr(a16, a16, 3, 1, a16, a16, 1, 3, 256, null) == 3                                # ratio 3
    assert lib.nhmc_data_sr(a16, a16, 4, 1, a16, a16, 1, 3, 250, null) == 3                                # dim % 4
    assert lib.nhmc_spectral_apply(a16, a16, a16, a16, a16, a16, a16, a16, 1, 3, 48, null) == 3           # dim % 32
    assert lib.nhmc_sum_partials(a16, 4, 2, 2, 1, a16, null) == 1                                          # offset >= stride
    assert lib.nhmc_metropolis(null, a16, a16, null, a16, null, 1, null) == 1
    # round-2 entry points
    assert lib.nhmc_leapfrog_first(a16, a16, a16, a16, null, a16, a16, 1.0, 1, 1024, a16, null) == 1       # x_out aliases x_in
    assert lib.nhmc_leapfrog_first(a16, P(0x2000), a16, a16, null, a16, a16, 1.0, 1, 1024, null, null) == 1  # needs sums_ws
    assert lib.nhmc_vq_nearest(a16, a16, a16, null, 1, 5, 4096, 8192, null) == 3                           # embed_dim 5
    assert lib.nhmc_vq_nearest(a16, null, a16, null, 1, 3, 4096, 8192, null) == 1
    assert lib.nhmc_gn_act_fwd(a16, a16, a16, null, 0, null, 0, 1e-5, 1, a16, a16, 1, 1, 64, 32, 9, null) == 3   # hw % 4
    assert lib.nhmc_gn_act_fwd(a16, a16, a16, null, 0, null, 0, 1e-5, 1, a16, a16, 1, 1, 48, 32, 64, null) == 3  # C % G
    assert lib.nhmc_gn_act_bwd(a16, null, a16, a16, null, 0, null, 0, 1e-5, 1, a16, null, a16, a16, 1, 1, 64, 32, 64, null) == 1
    assert lib.nhmc_gn_act_bwd(a16, a16, a16, a16, null, 0, null, 0, 1e-5, 1, a16, a16, a16, a16, 1, 1, 64, 32, 64, null) == 1   # dx_add aliases dx
    assert lib.nhmc_bias_add2(a16, a16, a16, a16, 1, 8, 6, null) == 3                                      # hw % 4
    assert lib.nhmc_ddim_mix_bwd_inpaint_px(a16, a16, 6, a16, a16, a16, a16, a16, 10, a16, a16, 0, a16, 1, 3, 100, null) == 3  # hw % 32
    # round-3 entry points
    assert lib.nhmc_ddim_mix_bwd_sr(a16, a16, 6, a16, a16, a16, 3, a16, a16, a16, 1, 3, 256, null) == 3                    # ratio 3
    assert lib.nhmc_sr_vjp_tiles(3, 256, 4) == 3 * 64 and lib.nhmc_sr_vjp_tiles(3, 256, 16) == lib.nhmc_sr_tiles(3, 256, 16)
    assert lib.nhmc_leapfrog_first_cached(a16, P(0x2000), a16, a16, a16, 512, a16, a16, 1.0, 1, 1024, a16, null) == 1     # pair_stride < n_chains * n_elem
    assert lib.nhmc_leapfrog_first_cached(a16, P(0x2000), a16, a16, null, 1024, a16, a16, 1.0, 1, 1024, a16, null) == 1    # no sel
    assert lib.nhmc_leapfrog_last_cached(a16, a16, a16, null, a16, a16, 1024, null, a16, 1, a16, a16, 1.0, 1, 1024, a16, null) == 1  # no loss
    assert lib.nhmc_grad_cache_store(a16, null, a16, a16, a16, a16, 2, 1024, 1, 1, 1024, null) == 1                        # flip not in {0, 1}
    assert lib.nhmc_grad_cache_store(a16, null, a16, a4, a16, a16, 0, 1024, 1, 1, 1024, null) == 2                         # alignment
    assert lib.nhmc_grad_cache_flip(null, a16, 1, null) == 1
    assert lib.nhmc_hamiltonian_cached(a16, 1, a16, null, 1, a16, 1.0, a16, null, 1, null) == 1                            # no sel
    assert lib.nhmc_sandwich_rect(a16, a16, a16, null, a16, a16, 1, 64, 64, 48, 64, null) == 3                             # C1 % 32
    assert lib.nhmc_data_spectral(a16, a16, a16, a16, null, 1, a16, a16, a16, 1, 3, 64, null) == 1                          # no transposed multiplier map
    assert lib.nhmc_data_spectral_vjp(a16, null, a16, a16, a16, a16, a16, 6, a16, a16, a16, a16, a16, a16, 1, 3, 64, null) == 1  # no transposed observation
    assert lib.nhmc_data_srconv(a16, a16, a16, a16, a16, a16, null, 1, a16, a16, a16, 1, 3, 64, 32, null) == 1              # no multiplier map
    assert lib.nhmc_data_srconv(a16, a16, a16, a16, a16, a16, a16, 1, a16, a16, a16, 1, 3, 64, 128, null) == 3             # small_dim > dim
    assert lib.nhmc_mass_from_variance(a16, 5, null, null, a16, a16, a16, a16, 1 << 20, 1, 768, null) == 1                 # no mass table
    assert lib.nhmc_inpaint_px_tiles(3, 65536) == 3 * 64 and lib.nhmc_gn_splits(32, 128, 32, 65536) >= 1
    assert lib.nhmc_spectral_project(a16, a16, a16, a16, a16, 1, 3, 48, null) == 3                         # dim % 32
    assert lib.nhmc_spectral_project(a16, null, a16, a16, a16, 1, 3, 64, null) == 1
    assert lib.nhmc_data_spectral_proj(a16, a16, a16, a16, 1, a16, a16, a16, 1, 3, 48, null) == 3
    assert lib.nhmc_data_spectral_proj(a16, null, a16, a16, 1, a16, a16, a16, 1, 3, 64, null) == 1          # needs y_proj
    assert lib.nhmc_data_spectral_proj_vjp(a16, a16, a16, a16, a16, a16, 5, a16, a16, a16, a16, a16, a16, 1, 3, 64, null) == 3  # e_channels
    assert lib.nhmc_data_spectral_proj_vjp(a16, a16, a16, a16, a16, a16, 6, a16, a16, P(0x1004), a16, a16, a16, 1, 3, 64, null) == 2  # alignment
    w5 = (ctypes.c_float * 5)(0.5, 0.5, 0.5, 0.0, -1.0)
    assert lib.nhmc_color_Ht(a16, ctypes.cast(w5, P), 1, a16, 1, 3, 64, null) == 3                         # H^+ with a zero singular value
    assert lib.nhmc_color_Ht(a16, ctypes.cast(w5, P), 0, a16, 1, 5, 64, null) == 3                         # channels > 4


def test_no_cpu_fallback_in_the_python_binding():
    import nhmc.kernels as K
    from nhmc._lib import NhmcError
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(NhmcError, match='no CPU path'):
        K.leapfrog_fused(K.LF_MID, x, x.clone(), x.clone(), 0.1, 1.0, 1.0)
    with pytest.raises(NhmcError, match='no CPU path'):
        K.ddim_mix_fwd(x, torch.zeros(1, 6, 8, 8), torch.ones(1), torch.ones(1))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import nhmc._lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'libnhmc.so'))
    with pytest.raises(L.NhmcError, match='has not been built'):
        L.load()


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'noise-space-hmc_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f'{f} imports the oracle'
