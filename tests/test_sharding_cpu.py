"""CPU: chain sharding logic and the single end-of-run gather (gloo, world_size 2 and 3)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import philox_ref


def test_chain_range_partitions_every_chain_exactly_once():
    from nhmc.sharding import chain_range, owner_of
    for n in (0, 1, 5, 64, 65, 512):
        for world in (1, 2, 3, 4, 8):
            spans = [chain_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            for c in range(0, n, max(1, n // 7)):
                lo, hi = spans[owner_of(c, n, world)]
                assert lo <= c < hi
    with pytest.raises(ValueError):
        chain_range(4, 2, 2)


def test_noise_is_keyed_by_global_chain_id_not_by_shard():
    """What makes results independent of the number of ranks: chain c's stream is (seed, c, draw)."""
    seed, n = 5678, 64
    whole = np.stack([philox_ref.randn_chain(seed, c, 4, n) for c in range(6)])
    from nhmc.sharding import chain_range
    for world in (2, 3):
        parts = []
        for r in range(world):
            lo, hi = chain_range(6, r, world)
            parts.append(np.stack([philox_ref.randn_chain(seed, lo + k, 4, n) for k in range(hi - lo)]))
        assert np.array_equal(np.concatenate(parts), whole)
    assert not np.array_equal(whole[0], whole[1])
    assert philox_ref.uniform_chain(seed, 3, 4) != philox_ref.uniform_chain(seed, 4, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_chains, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from nhmc import sharding
    r, _, w = sharding.init_process_group('gloo')
    assert (r, w) == (rank, world)
    lo, hi = sharding.chain_range(n_chains, rank, world)
    # per-chain "results": a scalar table [B_local, 3] and a sample block [B_local, 2, 4]
    stats = torch.stack([torch.tensor([c, c * c, rank], dtype=torch.float32) for c in range(lo, hi)]) if hi > lo \
        else torch.zeros(0, 3)
    samples = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 2, 4).contiguous()
    all_stats = sharding.gather_chains(stats, n_chains)
    all_samples = sharding.gather_chains(samples, n_chains)
    t = sharding.max_over_ranks(1.0 + rank, torch.device('cpu'))
    sharding.barrier()
    torch.save(dict(stats=all_stats, samples=all_samples, tmax=t), os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_chains', [(2, 6), (2, 5), (3, 4), (8, 512), (8, 13)])       # 8 ranks: the node's width (configs[2])
def test_gather_chains_over_gloo(tmp_path, world, n_chains):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_chains, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(tmp_path / f'r{r}.pt', weights_only=True)
        assert got['stats'].shape == (n_chains, 3)
        assert got['stats'][:, 0].tolist() == list(range(n_chains))                 # global chain order
        assert got['stats'][:, 1].tolist() == [c * c for c in range(n_chains)]
        assert got['samples'][:, 0, 0].tolist() == list(range(n_chains))
        assert got['tmax'] == float(world)                                            # max over ranks
