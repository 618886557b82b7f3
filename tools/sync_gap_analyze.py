#!/usr/bin/env python3
"""idle time of the GPU inside the last steps of a rocprofv3 kernel trace (tools/sync_gap_probe.py): gaps between the end of a
kernel and the start of the next one that begins after it, summed per step and listed when long."""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# steps end with the optimizer kernel
ends = [i for i, r in enumerate(rows) if 'sgd_momentum' in r[2]]
for s in range(len(ends) - 4, len(ends) - 1):
    seg = rows[ends[s] + 1: ends[s + 1] + 1]
    t0, busy_end, idle, gaps = rows[ends[s]][1], rows[ends[s]][1], 0, []
    for a, b, n in seg:
        if a > busy_end:
            idle += a - busy_end
            if a - busy_end > 30000:
                gaps.append((round((a - busy_end) / 1e3), n.split('(')[0][-50:]))
        busy_end = max(busy_end, b)
    print('step %d: %.2f ms from optimizer to optimizer, GPU idle %.2f ms; gaps > 30 us before: %s' % (s, (seg[-1][1] - t0) / 1e6, idle / 1e6, gaps[:12]))
