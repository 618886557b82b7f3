#!/usr/bin/env python3
"""Per-shape FETCH_SIZE / WRITE_SIZE of the GEMM launches of one tools/gemm_q_sweep.py child run under
`rocprofv3 --pmc <counter> --kernel-trace`: the child launches every shape (2 + GB_REPS) times in a row, so the
gemm256q dispatches are averaged in consecutive groups of that size.   pmc_gemm_groups.py <dir> <group size>"""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm256q' in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value']), r['Kernel_Name'][:60]))
rows.sort()
for i in range(0, len(rows), n):
    g = rows[i:i + n]
    mult = 2.0 if g[0][1] == 'FETCH_SIZE' else 1.0           # gfx950: FETCH_SIZE counts half the bytes
    print('%-14s %8.1f MB/launch (x%d)  %s' % (g[0][1], mult * sum(v for _, _, v, _ in g) / len(g) * 1024 / 1e6, len(g), g[0][3]))
