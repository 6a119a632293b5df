"""Condense the SQ counter passes of tools/profile_stalls.sh into one row per kernel: mean counter value per dispatch.
    python3 tools/stall_table.py gpurun_out/prof_stalls_r03 r03   ->  profiles/r03_pair256_stalls.csv   (third argument: the name part, default pair256)"""
import collections
import csv
import glob
import os
import re
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
part = sys.argv[3] if len(sys.argv) > 3 else 'pair256'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sorted(glob.glob(os.path.join(out_dir, 'pass*', '**', '*counter_collection.csv'), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r'\(.*$', '', row['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))
            if not name.startswith(('k_pair256', 'k_sgemm', 'k_mix_bwd_sr', 'k_mix_bwd_inpaint', 'k_leapfrog', 'k_fwht', 'k_sr', 'k_data_inpaint', 'k_mix_bwd', 'k_color')):
                continue
            cell = acc[name][row['Counter_Name']]
            cell[0] += float(row['Counter_Value'])
            cell[1] += 1
counters = sorted({c for k in acc.values() for c in k})
dst = os.path.join(ROOT, 'profiles', f'{tag}_{part}_stalls.csv')
with open(dst, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dispatches'] + counters)
    for name in sorted(acc):
        n = max(v[1] for v in acc[name].values())
        w.writerow([name, n] + [f'{acc[name][c][0] / acc[name][c][1]:.1f}' if acc[name][c][1] else '' for c in counters])
print('wrote', dst, len(acc), 'kernels', len(counters), 'counters')
