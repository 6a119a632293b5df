"""GPU box (one MI355X): the N > 1 control flow with two ranks sharing the device over gloo (NHMC_DIST_BACKEND=gloo,
NHMC_SHARED_GPU=1) -- the real CLI and the real sampler: per-image results equal the single-process run; and bench.py's
two-rank path (barriers, max-over-ranks timing, final gather) end to end with a small U-Net.  RCCL with N > 1 needs the
driver's multi-GPU node; tests/test_rccl_gpu.py covers the RCCL calls in a one-rank group."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _torchrun(n, script, args, cwd, extra_env=None):
    env = dict(os.environ, NHMC_DIST_BACKEND='gloo', NHMC_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), script] + args
    return subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


def _tiny_config(tmp_path):
    cfg = {'data': {'dataset': 'tiny', 'image_size': 32, 'channels': 3, 'rescaled': True},
           'model': dict(image_size=32, num_channels=32, num_res_blocks=1, channel_mult='1,2', learn_sigma=True,
                         class_cond=False, use_checkpoint=False, attention_resolutions='16', num_heads=4,
                         num_head_channels=16, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
                         resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path=''),
           'diffusion': {'beta_schedule': 'linear', 'beta_start': 1e-4, 'beta_end': 0.02, 'num_diffusion_timesteps': 1000}}
    (tmp_path / 'configs').mkdir()
    (tmp_path / 'configs' / 'config_tiny.yml').write_text(yaml.safe_dump(cfg))


def test_cli_results_do_not_depend_on_the_number_of_ranks(tmp_path):
    _tiny_config(tmp_path)
    args = ['--dataset', 'tiny', '--algo', 'hmc', '--timesteps', '3', '--deg', 'sr4', '--sigma_0', '0.05', '-i', str(tmp_path / 'out'),
            '--tau', '0.1', '--epsilon', '0.05', '--synthetic', '4', '--philox', '--hmc_epochs', '3', '--hmc_sampling', '2']
    script = os.path.join(ROOT, 'main_sampling.py')
    # the same per-process batch composition on both sides (one image at a time): a different batch size may pick
    # another MIOpen convolution solver, whose 1e-6 rounding differences can flip an accept decision of the fp32 run
    one = subprocess.run([sys.executable, script] + args + ['--chains', '1'], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = _torchrun(2, script, args + ['--chains', '1'], tmp_path)
    assert two.returncode == 0, two.stderr[-2000:]
    rows = lambda out: re.findall(r'image (\d+): PSNR ([-0-9.naninf]+) \(std over samples ([-0-9.naninf]+)\)', out)
    assert len(rows(one.stdout)) == 4 and rows(one.stdout) == rows(two.stdout), (one.stdout, two.stdout)


def test_bench_two_rank_control_flow_on_the_shared_gpu(tmp_path):
    out = _torchrun(2, os.path.join(ROOT, 'bench.py'),
                    ['--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '4', '--chunk', '2', '--tiny-score',
                     '--rehearse-shared-gpu', '--no-cpu-baseline', '--roofline-launches', '8'], ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_chains'] == 8 and line['final_gather']['chains'] == 8
    assert line['value'] > 0 and line['scaling'] == 'weak' and 'REHEARSAL' in line['config']['workload']


def test_bench_ranks_halve_the_score_chunk_together(tmp_path):
    """An out-of-memory error on ONE rank (injected on rank 1) makes EVERY rank retry with half the score chunk, so the
    barriers of the timed region still line up (bench.settle_score_chunk)."""
    out = _torchrun(2, os.path.join(ROOT, 'bench.py'),
                    ['--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '4', '--chunk', '4', '--tiny-score',
                     '--rehearse-shared-gpu', '--no-cpu-baseline', '--roofline-launches', '8'], ROOT,
                    extra_env={'NHMC_BENCH_FAKE_OOM': '1'})
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'every rank retries with --chunk 2' in out.stderr
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['config']['score_chunk'] == 2 and line['final_gather']['chains'] == 8 and line['value'] > 0


def _bare_bench(args, extra_env=None):
    """`python bench.py --gpus N ...` with NO launcher around it: bench.py must start its own ranks (as a child process)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=900)


@pytest.mark.parametrize('n', [2, 4])        # 4 ranks + this process = 5 processes on the card (the box allows 6)
def test_bare_bench_command_starts_its_own_ranks(n):
    out = _bare_bench(['--gpus', str(n), '--steps', '1', '--warmup', '1', '--batch', '2', '--chunk', '2', '--tiny-score',
                       '--rehearse-shared-gpu', '--no-cpu-baseline', '--roofline-launches', '8'])
    assert out.returncode == 0, out.stderr[-2000:]
    assert f'--gpus {n} without WORLD_SIZE: launching' in out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                               # rank 0 prints the one JSON line
    line = json.loads(lines[0])
    assert line['n_gpus'] == n and line['config']['global_chains'] == 2 * n and line['final_gather']['chains'] == 2 * n
    assert line['value'] > 0 and line['scaling'] == 'weak'


def test_bare_bench_command_hands_back_the_ranks_exit_code():
    out = _bare_bench(['--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '2', '--chunk', '1', '--tiny-score',
                       '--rehearse-shared-gpu', '--no-cpu-baseline', '--roofline-launches', '8', '--deg', 'no_such_degradation'])
    assert out.returncode != 0
