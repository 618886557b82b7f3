#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter group) into the per-kernel summary bench.py
reads: {kernel name: {counter: {"launches": n, "launches_per_step": n / steps, "avg_KB" | "avg": mean per launch}}}.

    python tools/pmc_summary.py [--steps N] profiles/pmc/r03_pmc_whole_step_summary.json gpurun_out/pmc_fetch/**/*.csv ...

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (FETCH_SIZE at half the bytes on gfx950: bench.py doubles it).
When a pass holds SQ_VALU_MFMA_BUSY_CYCLES and another GRBM_GUI_ACTIVE, every kernel also gets
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
(the busy counter is in per-SIMD cycles summed over the SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs:
MI355X_MICROARCH.md, cycle constants and the DVFS note), and eff_clock_GHz = GRBM_GUI_ACTIVE / 8 / kernel time when the
kernel-trace CSV of the same pass is given (files ending in kernel_trace.csv).
"""
import csv
import json
import sys

SIZES = ('FETCH_SIZE', 'WRITE_SIZE')


def main(out, files, steps):
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
    res = {}
    for kn, cs in acc.items():
        r = res.setdefault(kn, {})
        for c, v in cs.items():
            e = {'launches': v[0], ('avg_KB' if c in SIZES else 'avg'): v[1] / v[0]}
            if steps:
                e['launches_per_step'] = v[0] / steps
            r[c] = e
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
            busy = cs['SQ_VALU_MFMA_BUSY_CYCLES'][1] / cs['SQ_VALU_MFMA_BUSY_CYCLES'][0]
            act = cs['GRBM_GUI_ACTIVE'][1] / cs['GRBM_GUI_ACTIVE'][0]
            if act > 0:
                r['mfma_busy_frac'] = round(busy / (act / 8.0 * 256 * 4), 4)
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('%d kernels -> %s' % (len(res), out))


if __name__ == '__main__':
    args = sys.argv[1:]
    steps = 0
    if args and args[0] == '--steps':
        steps = int(args[1])
        args = args[2:]
    main(args[0], args[1:], steps)
