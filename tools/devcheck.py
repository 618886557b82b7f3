#!/usr/bin/env python3
"""Run every GPU parity check and print a table (does not stop at the first failure)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch  # noqa: E402

import gpu_checks  # noqa: E402

only = sys.argv[1] if len(sys.argv) > 1 else ''
bad = 0
for name, fn in gpu_checks.all_checks():
    if only and only not in name:
        continue
    t = time.time()
    try:
        err, tol = fn()
        torch.cuda.synchronize()
        ok = err <= tol and err == err
        bad += not ok
        print('%-44s err %.3e tol %.1e %s (%.2fs)' % (name, err, tol, 'ok' if ok else 'FAIL', time.time() - t), flush=True)
        if name.startswith('stem_oracle'):
            print('    worst:', ['%s=%.2e' % kv for kv in gpu_checks.stem_vs_oracle.last], flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print('%-44s EXC %s' % (name, repr(e)[:300]), flush=True)
        traceback.print_exc(limit=2)
print('failures:', bad)
