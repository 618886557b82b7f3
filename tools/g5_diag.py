"""G5 diagnosis: the HIP float32 model against the reference's float32 and float64 captures (logit, loss, per-tensor
gradient-norm ratios)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import istvt_pkg
import recipe

istvt_pkg.load()
from istvt_amd.network.vivit.vivit import XceptionVidTr

g = np.load(os.path.join(ROOT, 'tests/golden/G5_native.npz'))
h = np.load(os.path.join(ROOT, 'tests/golden/G5b_native_fp64.npz'))
model = XceptionVidTr()
sd = model.state_dict()
model.load_state_dict({k: torch.from_numpy(recipe.param_value(k, tuple(v.shape))) for k, v in sd.items()})
model = model.cuda().train()
x = torch.from_numpy(recipe.input_value('g5.x', (1, 6, 3, 300, 300))).cuda()
logits = model(x)
loss = torch.nn.BCEWithLogitsLoss()(logits.view(-1), torch.ones(1, device='cuda'))
loss.backward()
print('logit hip %.7f ref32 %.7f ref64 %.7f' % (float(logits), float(g['logits'].reshape(-1)[0]), float(h['logits64'].reshape(-1)[0])))
print('loss  hip %.8f ref32 %.8f ref64 %.8f' % (float(loss), float(g['loss']), float(h['loss64'])))
named = dict(model.named_parameters())
rows = []
for k in g['live_param_names']:
    k = str(k)
    got, r32, r64 = float(named[k].grad.norm()), float(g['gnorm.' + k]), float(h['gnorm64.' + k])
    rows.append((got / r64 - 1, r32 / r64 - 1, k))
a = np.array([r[0] for r in rows]); b = np.array([r[1] for r in rows])
print('hip/ref64 - 1: median %.3e mean %.3e min %.3e max %.3e' % (np.median(a), a.mean(), a.min(), a.max()))
print('ref32/ref64 - 1: median %.3e mean %.3e min %.3e max %.3e' % (np.median(b), b.mean(), b.min(), b.max()))
for r in sorted(rows)[:12] + sorted(rows)[-6:]:
    print('%+.3e  (ref32 %+.3e)  %s' % r)
for k in g.files:
    if k.startswith('grad.'):
        got = named[k[5:]].grad.reshape(-1)[:64].double().cpu().numpy()
        r32, r64 = g[k].astype(np.float64), h['grad64.' + k[5:]]
        print(k, 'hip-ref64 %.3e  ref32-ref64 %.3e  |ref64| %.3e' % (np.linalg.norm(got - r64), np.linalg.norm(r32 - r64), np.linalg.norm(r64)))
