"""CPU, world_size 2 over gloo: the CLI's image sharding, its per-image keyed inputs and the single end-of-run gather
give the same per-image table as one process (the sampler itself needs a GPU; a deterministic stand-in takes its place --
tests/test_multirank_gpu.py runs the real one with two ranks on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_IMAGES, M, SHAPE, SEED = 7, 11, (3, 4, 4), 5678


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(rank, world, chains):
    """What cli.main does per rank, with a stand-in for sampler.hmc: rows of [image id, f(y_0, x), chain_id0]."""
    from nhmc import cli, sharding
    g = torch.Generator().manual_seed(1)
    y_clean = torch.randn(N_IMAGES, M, generator=g)                 # H(x_orig) of every image
    rows = []
    for batch in cli.image_batches(N_IMAGES, rank, world, chains):
        drawn = [cli.draw_inputs(SEED, s, y_clean[s], 0.1, SHAPE) for s in batch]
        for s, (y0, x) in zip(batch, drawn):
            rows.append([float(s), float(y0.double().sum() + 3 * x.double().sum()), float(y0[0])])
    local = torch.tensor(rows, dtype=torch.float32).reshape(-1, 3)
    return sharding.gather_chains(local, N_IMAGES, rank, world)


def _worker(rank, world, port, chains, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from nhmc import sharding
    sharding.init_process_group('gloo')
    table = _run(rank, world, chains)
    sharding.barrier()
    torch.save(table, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_image_inputs_do_not_depend_on_sharding_or_chains(tmp_path):
    os.environ.pop('WORLD_SIZE', None)
    single = _run(0, 1, 1)
    assert single[:, 0].tolist() == list(range(N_IMAGES))
    assert torch.equal(_run(0, 1, 3), single)                      # --chains changes the batching, not the draws
    for world, chains in ((2, 1), (2, 3), (3, 2)):
        port = _free_port()
        d = tmp_path / f'w{world}c{chains}'
        d.mkdir()
        mp.spawn(_worker, args=(world, port, chains, str(d)), nprocs=world, join=True)
        for r in range(world):
            assert torch.equal(torch.load(d / f'r{r}.pt', weights_only=True), single), (world, chains, r)


def test_image_batches_never_straddle_ranks():
    from nhmc import cli
    for n in (1, 5, 7, 64):
        for world in (1, 2, 3, 8):
            for chains in (1, 2, 16):
                seen = []
                for r in range(world):
                    for b in cli.image_batches(n, r, world, chains):
                        assert 1 <= len(b) <= chains and b == list(range(b[0], b[-1] + 1))
                        seen += b
                assert seen == list(range(n))
