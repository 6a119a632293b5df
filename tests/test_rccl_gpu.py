"""GPU: the RCCL calls of the sharded path (all_gather of padded per-chain rows, fp64 MAX all_reduce, barrier) run on
this ROCm build.  One GPU is what the test box has, so the group has ONE rank (backend "nccl" = RCCL); the N > 1
partition / padding logic is covered by the gloo tests in test_sharding_cpu.py.  Own process: the default process group
is per-process state."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    import torch, torch.distributed as dist
    from nhmc import sharding
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    assert dist.get_backend() == 'nccl'
    local = torch.arange(12, dtype=torch.float32, device='cuda').reshape(4, 3)
    parts = [torch.empty_like(local)]
    dist.all_gather(parts, local)                                   # the collective gather_chains issues
    assert torch.equal(parts[0], local)
    t = torch.tensor([1.25], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                        # what max_over_ranks issues at N > 1
    assert float(t.item()) == 1.25 and sharding.max_over_ranks(1.25, torch.device('cuda', 0)) == 1.25
    dist.barrier()
    out = sharding.gather_chains(local, 4)
    assert torch.equal(out, local)
    dist.destroy_process_group()
    print('rccl ok')
''') % ROOT


def test_rccl_collectives_of_the_sharded_path_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and 'rccl ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
