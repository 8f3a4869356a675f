#!/usr/bin/env python3
"""Timeline of the last LML evaluation in a rocprofv3 kernel trace (csv directory): per launch its start relative to the
evaluation's kernel-matrix launch, duration, queue and grid -- shows what runs beside the serial chain of panel kernels.
   python tools/fit_timeline.py <dir> [max rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
i0 = [i for i, r in enumerate(rows) if 'kmat' in r['Kernel_Name']][-1]
t0 = int(rows[i0]['Start_Timestamp'])
queues = {}
short = {'chol_panel_kernel': 'panel', 'gemm_f64_kernel<false, false': 'gemm NN(update)', 'gemm_f64_kernel<false, true': 'gemm NT(trtri)',
         'gemm_f64_kernel<true, true': 'gemm TT(WtW)'}
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
last_end = {}
for r in rows[i0:i0 + n]:
    name = r['Kernel_Name']
    for k, v in short.items():
        if k in name:
            name = v
    name = name.split('(')[0][-28:] if len(name) > 28 else name
    q = queues.setdefault(r['Queue_Id'], len(queues))
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = s - last_end.get(q, s)
    last_end[q] = e
    grid = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
    wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '1'))
    try:
        nwg = int(grid) // max(int(wg), 1)
    except ValueError:
        nwg = grid
    print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  gap {gap / 1e3:6.1f}  wgs {nwg!s:>6}  {'    ' * q}{name}")
