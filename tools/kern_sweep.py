#!/usr/bin/env python3
"""Run bench.py under rocprofv3 for one env setting and print ms/step of the kernels whose name contains a pattern
(used for grid-size sweeps): python tools/kern_sweep.py <pattern> -- the env var is set by the caller."""
import csv, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
d = tempfile.mkdtemp(dir='/tmp')
subprocess.run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d, '-o', 'k', '--', 'python3',
                os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-kernel-profile',
                '--no-wgrad-overlap'], cwd='/tmp', stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
for root, _, files in os.walk(d):
    for f in files:
        if f.endswith('kernel_stats.csv'):
            for r in csv.DictReader(open(os.path.join(root, f))):
                if pat in r['Name']:
                    print('%-60s %6.1f/step %8.3f ms/step' % (r['Name'][:60], int(r['Calls']) / 7, float(r['TotalDurationNs']) / 7e6), flush=True)
