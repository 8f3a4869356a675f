#!/usr/bin/env python3
"""Durations of the chol_panel_kernel launches of the last evaluation in a rocprofv3 kernel trace (csv directory)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
i0 = [i for i, r in enumerate(rows) if 'kmat' in r['Kernel_Name']][-1]
print([round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1) for r in rows[i0:i0 + 60] if 'chol_panel' in r['Kernel_Name']])
