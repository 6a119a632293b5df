"""CPU: the command-line contract of bench.py the driver depends on (flags, defaults, the keys of the one JSON line),
checked without a GPU; the numbers themselves are produced on the MI355X box."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flags_and_defaults(monkeypatch):
    import bench
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse()
    assert (a.gpus, a.batch or bench.B_PER_GPU, a.deg, a.latent) == (1, 64, 'inpaint_random', False) and 1 <= a.steps <= 10 and a.warmup >= 1
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '5', '--warmup', '2'])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--latent'])
    a = bench.parse()
    assert a.latent and (a.batch or bench.B_LATENT) == 16                        # configs[4]: 128 chains over 8 GPUs


def test_json_line_carries_the_contract_keys():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert re.search(r"'%s'\s*:" % key, src), key
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):          # roofline object
        assert re.search(r"\b%s=" % key, src), key
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):                       # cpu_baseline object
        assert re.search(r"\b%s=" % key, src), key
    assert "'scaling': 'weak'" in src and "'vs_baseline': None" in src and "'dtype': 'f32'" in src


def test_traffic_file_is_what_bench_reads():
    t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_leapfrog.json')))
    alg = 5 * 64 * 3 * 256 * 256 * 4                                               # 5T x 64 chains
    assert abs(t['hbm_bytes_per_launch'] - alg) / alg < 0.01                       # PMC traffic = algorithmic bytes
    assert t['kernel'].startswith('k_leapfrog<1, false')


def test_only_the_cpu_baseline_leg_touches_the_oracle():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    uses = [m.start() for m in re.finditer(r'\boracle\b', src)]
    lo, hi = src.index('def cpu_baseline'), src.index('def single_chain_rate')
    code_uses = [u for u in uses if 'import' in src[src.rfind('\n', 0, u):src.find('\n', u)]]
    assert code_uses and all(lo < u < hi for u in code_uses)
