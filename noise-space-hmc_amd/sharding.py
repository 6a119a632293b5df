"""Multi-GPU: shard chains, never data.  One process per GPU, `torch.distributed` ("nccl" = RCCL on
ROCm, xGMI between the GPUs of a node; "gloo" in the CPU tests).

Chains are independent units (the score network has no cross-sample op; loss, Hamiltonian, accept and
schedules are per chain), so the hot loop has NO collective.  Global chains are split into W contiguous
blocks whose sizes differ by at most one (`chain_range`: the first B mod W ranks take one more); a chain's
noise is keyed by its global id (PhiloxNoise) and, in the CLI, so are its measurement noise and start
point, so the result does not depend on W.  The only exchange is one all_gather at the end: per-chain
scalars (and, if asked, the collected samples).

Rehearsal on a one-GPU box (tests): NHMC_DIST_BACKEND=gloo selects the backend and NHMC_SHARED_GPU=1 puts
every rank on cuda:0; `gather_chains` then stages device tensors through the host for the collective.
The reference has no multi-device path at all (`device_count` read and unused,
main_sampling.py:369).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    local = 0 if os.environ.get('NHMC_SHARED_GPU') == '1' else int(os.environ.get('LOCAL_RANK', 0))
    return int(os.environ.get('RANK', 0)), local, int(os.environ.get('WORLD_SIZE', 1))


def init_process_group(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get('NHMC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def chain_range(n_chains, rank, world):
    """Contiguous block partition of global chain ids: [lo, hi) for `rank`.  Blocks differ by at most
    one chain; empty ranges are allowed (more ranks than chains)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError('bad rank/world')
    base, extra = divmod(n_chains, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(chain, n_chains, world):
    for r in range(world):
        lo, hi = chain_range(n_chains, r, world)
        if lo <= chain < hi:
            return r
    raise ValueError('chain id out of range')


def gather_chains(local, n_chains, rank=None, world=None):
    """all_gather of a per-chain tensor [B_local, ...] -> [n_chains, ...] on every rank, in global chain
    order.  Ragged shards are padded to the largest shard for the collective and trimmed after."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return local
    sizes = [chain_range(n_chains, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    via_host = local.is_cuda and dist.get_backend() == 'gloo'           # shared-GPU rehearsal: gloo moves host memory
    src = local.cpu() if via_host else local
    pad = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    return out.to(local.device) if via_host else out


def max_over_ranks(value, device):
    """Scalar max over ranks (bench timing contract)."""
    if dist.is_initialized() and dist.get_backend() == 'gloo':
        device = torch.device('cpu')
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
