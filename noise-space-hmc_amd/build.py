"""Build libnhmc.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python noise-space-hmc_amd/build.py [--force]

The shared object lands next to the sources' package (noise-space-hmc_amd/libnhmc.so); it is
git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc cross-compiles for
gfx950 without a GPU, so this runs in the build container and on the box alike.
"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libnhmc.so')
ARCH = 'gfx950'

# -ffp-contract=off: the elementwise kernels reproduce the reference's separate fp32 mul/add ops
# (ATen issues them unfused); the MFMA GEMM file is unaffected (explicit intrinsics).
FLAGS = ['-O3', '-std=c++17', f'--offload-arch={ARCH}', '-fPIC', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-function']
OBJ = os.path.join(PKG, 'build')          # per-source objects (git- and gpurun-ignored): only changed files recompile


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(os.path.dirname(PKG), 'include', 'nhmc.h'), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm with gfx950 support)')
    return exe


def _obj_stale(src, obj, common):
    return not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in [src] + common)


def build(force=False, verbose=True, jobs=4):
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    common = glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(os.path.dirname(PKG), 'include', 'nhmc.h'),
                                                     os.path.abspath(__file__)]
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _obj_stale(src, obj, common):
            todo.append([hipcc()] + FLAGS + ['-c', src, '-o', obj])
    running = []
    while todo or running:
        while todo and len(running) < jobs:
            cmd = todo.pop(0)
            if verbose:
                print(' '.join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, proc = running.pop(0)
        if proc.wait() != 0:
            for _, other in running:
                other.wait()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}'] + objs + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
