"""GPU: a plain C++ program (tools/abi_demo.cpp) consumes libnhmc.so through include/nhmc.h -- no Python, no torch."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_consumer_of_the_c_abi(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    exe = str(tmp_path / 'abi_demo')
    lib_dir = os.path.join(ROOT, 'noise-space-hmc_amd')
    subprocess.run([hipcc, '-O2', '-ffp-contract=off', os.path.join(ROOT, 'tools', 'abi_demo.cpp'), '-I' + os.path.join(ROOT, 'include'),
                    '-L' + lib_dir, '-lnhmc', '-Wl,-rpath,' + lib_dir, '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert '0 of' in out.stdout and 'ABI demo OK' in out.stdout
