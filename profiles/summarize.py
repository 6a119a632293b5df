#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_rNN/...) into the small files kept under profiles/.

    python profiles/summarize.py gpurun_out/prof_r01 r01

Writes profiles/<tag>_kernel_stats_hip.csv   (our HIP kernels, from the --kernel-only run)
       profiles/<tag>_kernel_stats_e2e_top.csv (top 40 kernels of the end-to-end run)
       profiles/<tag>_pmc_hbm.csv            (per-kernel mean FETCH_SIZE / WRITE_SIZE per launch)
       profiles/traffic_leapfrog.json        (HBM bytes per launch of the dominant kernel; read by bench.py)
Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
streaming read, so it is doubled for our 16-B-per-lane kernels.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
OURS = ('k_leapfrog', 'k_mix_', 'k_map_back', 'k_vq_', 'k_gn_', 'k_pair', 'k_data_inpaint', 'k_inpaint', 'k_sr', 'k_sgemm', 'k_sum_partials',
        'k_hamiltonian', 'k_metropolis', 'k_schedule', 'k_accept_commit', 'k_psnr', 'k_randn', 'k_uniform',
        'k_color', 'k_fwht', 'k_cs_', 'k_copy_probe', 'k_latent', 'k_mass', 'k_rank')


def short(name):
    n = name.replace('void ', '').replace('(anonymous namespace)::', '')
    return n.split('(')[0]


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    return hits[0] if hits else None


def stats(path, out, keep):
    rows = list(csv.DictReader(open(path)))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        for r in rows:
            if keep(r):
                w.writerow([short(r['Name'])[:110], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'],
                            r['MinNs'], r['MaxNs'], r['StdDev']])


p = one('kern/*/*_kernel_stats.csv')
if p:
    stats(p, os.path.join(here, f'{tag}_kernel_stats_hip.csv'), lambda r: any(k in r['Name'] for k in OURS))
p = one('e2e/*/*_kernel_stats.csv')
if p:
    rows = list(csv.DictReader(open(p)))
    top = set(r['Name'] for r in rows[:40])
    stats(p, os.path.join(here, f'{tag}_kernel_stats_e2e_top.csv'), lambda r: r['Name'] in top or any(k in r['Name'] for k in OURS))

# steady-state table of the end-to-end run: only the dispatches between bench.py's two marker launches
# (NHMC_PROFILE_MARK=1: one-block k_copy_probe right before and after the timed steps), i.e. no warm-up, no MIOpen
# first-call fallbacks, no roofline / hot-path legs.  Run on the box (the trace is tens of MB); keeps a small csv.
p = one('e2e/*/*_kernel_trace.csv')
if p:
    rows = list(csv.DictReader(open(p)))
    gx = 'Grid_Size_X' if 'Grid_Size_X' in rows[0] else 'Grid_Size'
    marks = [r for r in rows if 'k_copy_probe' in r['Kernel_Name'] and int(r[gx]) == 256]
    if len(marks) >= 2:
        t_lo, t_hi = int(marks[0]['End_Timestamp']), int(marks[1]['Start_Timestamp'])
        steps = int(os.environ.get('NHMC_PROFILE_STEPS', '2'))
        agg = defaultdict(lambda: [0, 0])

        def family(n):
            if any(k in n for k in OURS):
                return 'nhmc HIP kernels'
            low = n.lower()
            if 'cijk_' in low or 'gemm' in low and 'conv' not in low and 'igemm' not in low:
                return 'GEMM (rocBLAS / hipBLASLt: attention, linear)'
            if any(k in low for k in ('conv', 'igemm', 'winograd', 'sp3asm', 'naive_')):
                return 'MIOpen convolution'
            if any(k in low for k in ('groupnorm', 'rowwisemoments', 'computefusedparams', 'group_norm', 'gammabeta', 'batch_norm', 'batchnorm')):
                return 'GroupNorm (ATen)'
            if 'softmax' in low:
                return 'softmax (ATen)'
            if any(k in low for k in ('elementwise', 'vectorized', 'cat', 'copy', 'fill', 'upsample', 'avg_pool', 'reduce', 'index')):
                return 'ATen elementwise / copy / pool'
            return 'other'
        for r in rows:
            st = int(r['Start_Timestamp'])
            if t_lo <= st <= t_hi:
                a = agg[short(r['Kernel_Name'])[:110]]
                a[0] += 1
                a[1] += int(r['End_Timestamp']) - st
        total = sum(v[1] for v in agg.values())
        fam = defaultdict(int)
        for k, v in agg.items():
            fam[family(k)] += v[1]
        with open(os.path.join(here, f'{tag}_kernel_stats_e2e_steady.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow([f'# steady state: {steps} timed steps of bench.py (B = 64, U-Net in the loop), wall between markers '
                        f'{(t_hi - t_lo) / 1e6:.1f} ms, summed kernel time {total / 1e6:.1f} ms'])
            w.writerow(['family', 'ms_per_step', 'percent'])
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
                w.writerow([k, f'{v / 1e6 / steps:.2f}', f'{100 * v / total:.2f}'])
            w.writerow([])
            w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
                w.writerow([k, v[0], v[1], f'{v[1] / v[0]:.0f}', f'{100 * v[1] / total:.3f}'])
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[45:]:
                if any(o in k for o in OURS):
                    w.writerow([k, v[0], v[1], f'{v[1] / v[0]:.0f}', f'{100 * v[1] / total:.3f}'])

acc = defaultdict(lambda: defaultdict(list))
for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    p = one(f'{sub}/*/*_counter_collection.csv')
    if not p:
        continue
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in OURS):
            acc[short(r['Kernel_Name'])][counter].append(float(r['Counter_Value']))
if acc:
    with open(os.path.join(here, f'{tag}_pmc_hbm.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches', 'FETCH_SIZE_KiB_mean', 'WRITE_SIZE_KiB_mean', 'hbm_bytes_per_launch(2*FETCH+WRITE)'])
        for k, d in sorted(acc.items()):
            fe = sum(d['FETCH_SIZE']) / max(1, len(d['FETCH_SIZE']))
            wr = sum(d['WRITE_SIZE']) / max(1, len(d['WRITE_SIZE']))
            w.writerow([k[:110], len(d['FETCH_SIZE']), f'{fe:.1f}', f'{wr:.1f}', f'{(2 * fe + wr) * 1024:.0f}'])
            if k.startswith('k_leapfrog<1, false'):
                json.dump({'kernel': k, 'fetch_size_kib': fe, 'write_size_kib': wr,
                           'hbm_bytes_per_launch': (2 * fe + wr) * 1024,
                           'source': f'profiles/{tag}_pmc_hbm.csv',
                           'note': 'FETCH_SIZE doubled (gfx950 wide-read correction), separate --pmc passes, source ' + tag},
                          open(os.path.join(here, 'traffic_leapfrog.json'), 'w'), indent=1)

# operator pass (tools/profile_ops.sh -> gpurun_out/prof_ops_rNN): python profiles/summarize.py gpurun_out/prof_ops_r01 r01
p = one('stats/*/*_kernel_stats.csv')
if p:
    stats(p, os.path.join(here, f'{tag}_kernel_stats_ops.csv'), lambda r: any(k in r['Name'] for k in OURS))
cols = ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES']
ops = defaultdict(lambda: defaultdict(list))
for sub in ('pmc_lds', 'pmc_mfma'):
    p = one(f'{sub}/*/*_counter_collection.csv')
    if not p:
        continue
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] in cols and any(k in r['Kernel_Name'] for k in OURS):
            ops[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
if ops:
    with open(os.path.join(here, f'{tag}_pmc_ops.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel (tools/ops_bench.py 16: B=16, 3x256x256)'] + cols)
        for k, d in sorted(ops.items()):
            w.writerow([k[:110]] + [f'{sum(d[c]) / len(d[c]):.0f}' if d[c] else '' for c in cols])
print('ok')
