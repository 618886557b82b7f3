import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import gpu_checks as G
for dt in ():
    for args in ((2, 33), (3, 77)):
        print('convdense', dt, args, G.conv_dense_check(dt, *args), flush=True)
for side in (96, 139):
    print('stem f32', side, G.stem_vs_oracle(torch.float32, side), G.stem_vs_oracle.last, flush=True)
print('stem bf16 139', G.stem_vs_oracle(torch.bfloat16, 139, init='random'), G.stem_vs_oracle.last, flush=True)
