#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter group) into the per-kernel summary bench.py
reads: {kernel name: {counter: {"launches": n, "avg_KB": mean value per launch}}}.

    python tools/pmc_summary.py profiles/pmc/r01_pmc_whole_step_summary.json gpurun_out/pmc_fetch/*.csv gpurun_out/pmc_write/*.csv

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (FETCH_SIZE at half the bytes on gfx950: bench.py doubles it).
"""
import csv
import json
import sys


def main(out, files):
    acc = {}
    for f in files:
        with open(f, newline='') as fh:
            rd = csv.DictReader(fh)
            if not rd.fieldnames or 'Counter_Name' not in rd.fieldnames:
                continue
            for row in rd:
                k = acc.setdefault(row['Kernel_Name'], {}).setdefault(row['Counter_Name'], [0, 0.0])
                k[0] += 1
                k[1] += float(row['Counter_Value'])
    res = {kn: {c: {'launches': v[0], 'avg_KB': v[1] / v[0]} for c, v in cs.items()} for kn, cs in acc.items()}
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('%d kernels -> %s' % (len(res), out))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
