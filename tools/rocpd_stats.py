#!/usr/bin/env python3
"""Export the per-kernel statistics of a rocprofv3 run (rocpd sqlite output, `*_results.db`) as the same CSV
`rocprofv3 --kernel-trace --stats --output-format csv` writes (`*_kernel_stats.csv`).

    python tools/rocpd_stats.py gpurun_out/prof_e/e_results.db profiles/r01_e_kernel_stats.csv
"""
import csv
import math
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    by = {}
    for name, dur in con.execute('select name, duration from kernels'):
        by.setdefault(name, []).append(dur)
    total = float(sum(sum(v) for v in by.values()))
    rows = []
    for name, v in by.items():
        n, s = len(v), float(sum(v))
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
        rows.append((name, n, int(s), round(mean, 6), round(100.0 * s / total, 2), min(v), max(v), round(sd, 6)))
    rows.sort(key=lambda r: -r[2])
    with open(out, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        w.writerows(rows)
    print('%d kernels, %.3f ms total -> %s' % (len(rows), total / 1e6, out))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
