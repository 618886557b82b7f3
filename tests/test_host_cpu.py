"""CPU-only checks: the C-ABI library loads and exports every symbol include/istvt_hip.h
declares, the nn.Module mirror keeps the reference's constructor surface and state-dict names,
the product path refuses to run without a GPU (no fallback), and the data-parallel gradient
bucket is correct across 2 gloo ranks."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pkg():
    import istvt_pkg
    return istvt_pkg.load()


def test_library_exports_every_declared_symbol(pkg):
    from istvt_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.lib()                                  # resolves all SIGNATURES or raises
    header = open(os.path.join(ROOT, 'include', 'istvt_hip.h')).read()
    declared = set(re.findall(r'\bint\s+(istvt_\w+)\s*\(', header))
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), 'libistvt_hip.so lacks %s declared in include/istvt_hip.h' % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_missing_library_fails_loudly(pkg, tmp_path, monkeypatch):
    from istvt_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.IstvtLibraryError):
        _lib.lib()


def test_no_cpu_fallback(pkg):
    from istvt_amd.network.vivit import module as M
    from istvt_amd.network.vivit.vivit import XceptionVidTr
    with pytest.raises(RuntimeError, match='ROCm device'):
        M.FeedForward(64, 128)(torch.zeros(1, 4, 64))
    with pytest.raises(RuntimeError, match='ROCm device'):
        XceptionVidTr(num_frames=2, grid=6, depth=1)(torch.zeros(1, 2, 3, 96, 96))


def test_product_code_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, '2023-tifs-istvt_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_state_dict_matches_reference(pkg, golden_dir):
    """names + shapes of XceptionVidTr().state_dict() equal the reference's (captured in G0)."""
    from istvt_amd.network.vivit.vivit import XceptionVidTr
    ref = json.load(open(os.path.join(golden_dir, 'G0_state_dict.json')))
    sd = XceptionVidTr().state_dict()
    assert list(sd.keys()) == [k for k, _ in ref['entries']]
    for k, shape in ref['entries']:
        assert list(sd[k].shape) == shape, k
    assert sum(p.numel() for p in XceptionVidTr().parameters()) == ref['num_parameters'] == 109172051


def test_constructor_surface(pkg):
    from istvt_amd.network.vivit import module as M, vivit as V
    from istvt_amd.network import xception as X
    from istvt_amd.network.models import model_selection, TransferModel
    V.DSTTr(19, 1, 1, 6, 728, 1, 8, 'cls', 728, 64, 0., 0., 4)            # positional, as vivit.py:104-105
    V.STTransformer(64, 1, 2, 32, 128, 0.)
    M.TemporalResidualAttention(64, 2, 32, 0.)
    M.SpatialOnlyAttention(64, 2, 32, 0.)
    M.FeedForward(64, 128, 0.)
    M.PreNorm(64, M.FeedForward(64, 128))
    X.SeparableConv2d(64, 128, 3, 1, 1, 1, False)
    X.Block(64, 128, 2, 2, False, True)
    assert X.Block(128, 256, 2, 2, True, True).rep[0].inplace is False          # skip sees the un-rectified input
    assert [n for n, _ in X.Block(64, 128, 2, 2, start_with_relu=False).rep.named_children()] == ['0', '1', '2', '3', '4', '5']
    m = model_selection('xception', 2, dropout=0.5, batch_size=1)
    assert isinstance(m, TransferModel) and hasattr(m, 'low_level_features')
    assert m.model.last_linear[1].out_features == 2
    assert type(model_selection('resnet_3d', 1, dropout=0.5, batch_size=4)).__name__ == 'XceptionVidTr'
    with pytest.raises(Exception, match='Choose valid model'):
        model_selection('resnet50', 1)


def test_live_parameters_and_bucket(pkg):
    from istvt_amd import parallel
    from istvt_amd.network.vivit.vivit import XceptionVidTr
    model = XceptionVidTr(num_frames=8, grid=14)
    live = parallel.live_named_parameters(model)
    n = sum(p.numel() for _, p in live)
    assert n == 89033873                     # 1,106,760 stem + 87,927,113 transformer
    small = XceptionVidTr(num_frames=2, grid=6, depth=1)
    ps = [p for _, p in parallel.live_named_parameters(small)]
    b = parallel.GradBucket(ps)
    assert all(p.grad.data_ptr() >= b.flat.data_ptr() for p in ps)
    ps[0].grad.add_(1.0)
    assert float(b.flat.sum()) == ps[0].numel()
    b.zero()
    assert float(b.flat.abs().sum()) == 0.0


_DP_WORKER = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import istvt_pkg; istvt_pkg.load()
from istvt_amd import parallel
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.manual_seed(1234 + rank)                      # deliberately different init per rank
lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 1))
parallel.broadcast_parameters(lin)                  # -> rank 0's weights everywhere
params = list(lin.parameters())
bucket = parallel.GradBucket(params)
g = torch.Generator().manual_seed(7)
x = torch.randn(8, 6, generator=g); y = (torch.rand(8, generator=g) > .5).float()
xs, ys = parallel.shard_batch(x, rank, world), parallel.shard_batch(y, rank, world)
bucket.zero()
loss = torch.nn.functional.binary_cross_entropy_with_logits(lin(xs).view(-1), ys)
loss.backward()
bucket.all_reduce(chunks=2)
# single-process reference on the concatenated batch with rank 0's weights
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 1))
ref.load_state_dict(lin.state_dict())
torch.nn.functional.binary_cross_entropy_with_logits(ref(x).view(-1), y).backward()
err = max(float((p.grad - q.grad).abs().max()) for p, q in zip(lin.parameters(), ref.parameters()))
# the same step with the all-reduce of the last layer's slice started early (the hook the token-assembly backward calls
# on the GPU is invoked by hand here, between "its gradients are complete" and the end of backward)
for p in params:
    p._istvt_fused_grad = True
bucket.enable_early_all_reduce(2)
bucket.zero()
torch.nn.functional.binary_cross_entropy_with_logits(lin(xs).view(-1), ys).backward()
bucket._on_ready(bucket.flat.device)
early_used = bucket._early_work is not None
bucket.all_reduce()
bucket.disable_early_all_reduce()
err_early = max(float((p.grad - q.grad).abs().max()) for p, q in zip(lin.parameters(), ref.parameters()))
err = max(err, err_early) if early_used else 1.0
w0 = [float(p.detach().sum()) for p in lin.parameters()]
out = torch.tensor(w0); gathered = [torch.zeros_like(out) for _ in range(world)]
dist.all_gather(gathered, out)
same = all(torch.equal(gathered[0], t) for t in gathered)
if rank == 0:
    print(json.dumps({'err': err, 'same_weights': same}))
dist.destroy_process_group()
'''


def test_data_parallel_two_ranks_gloo(tmp_path):
    """W-rank averaged gradients == 1-rank gradients on the concatenated batch."""
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29541', str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['same_weights']
    assert res['err'] < 1e-6


def test_force_collectives_in_a_world_of_one(tmp_path, pkg):
    """ISTVT_FORCE_COLLECTIVES=1 (what `bench.py --rccl-rehearsal` sets): a one-rank process group still goes through every
    collective of the data-parallel step -- early asynchronous slice, blocking rest, deferred 1/W -- and leaves the values
    as they were.  Run in a child process: the process group is global state."""
    script = tmp_path / 'force_worker.py'
    script.write_text("""
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
os.environ['ISTVT_FORCE_COLLECTIVES'] = '1'
import torch, torch.distributed as dist
import istvt_pkg; istvt_pkg.load()
from istvt_amd import parallel
dist.init_process_group('gloo', init_method='file://' + sys.argv[1], rank=0, world_size=1)
calls = []
real = dist.all_reduce
def counting(t, *a, **k):
    calls.append(t.numel())
    return real(t, *a, **k)
dist.all_reduce = counting
lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 1))
parallel.broadcast_parameters(lin)
params = list(lin.parameters())
bucket = parallel.GradBucket(params)
for p in params:
    p._istvt_fused_grad = True
bucket.zero()
lin(torch.ones(3, 6)).sum().backward()
before = bucket.flat.clone()
bucket.enable_early_all_reduce(2)
bucket._on_ready(bucket.flat.device)
early = bucket._early_work is not None
bucket.all_reduce()
print(json.dumps({'single': parallel._single_rank(), 'early': early, 'calls': calls,
                  'same': bool(torch.equal(before, bucket.flat)), 'numel': bucket.numel}))
dist.destroy_process_group()
""" % {'root': ROOT})
    r = subprocess.run([sys.executable, str(script), str(tmp_path / 'store')], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert res['single'] is False and res['early'] and res['same']
    assert res['calls'] == [6, 35] and res['numel'] == 41        # early: the last Linear's slice; then the first one's


def test_bench_cli_contract():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for flag in ('--gpus', '--steps', '--warmup'):
        assert flag in src
    for key in ('"metric"', 'roofline', 'cpu_baseline', 'ms_per_step', 'n_gpus', 'vs_baseline'):
        assert key.strip('"') in src


# ------------------------------------------------------------------------------------------------------------------ round 2
def test_pretrained_loader_unsqueezes_pointwise(pkg, tmp_path, monkeypatch):
    """reference xception.py:422-442: the ImageNet file stores the pointwise weights 2-D and the classifier as `fc`;
    return_pytorch04_xception(pretrained=True) unsqueezes them and renames fc -> last_linear.  TransferModel('xception')
    -- the reference's default call, models_copy.py:35 -- finds the file through $ISTVT_XCEPTION_WEIGHTS."""
    from istvt_amd.network import xception as X
    from istvt_amd.network.models import TransferModel
    torch.manual_seed(5)
    src = X.Xception()
    for p in src.parameters():
        torch.nn.init.uniform_(p, -1.0, 1.0)
    file_sd = {k: (v[:, :, 0, 0].clone() if 'pointwise' in k else v.clone()) for k, v in src.state_dict().items()}
    assert file_sd['block1.rep.0.pointwise.weight'].dim() == 2
    path = str(tmp_path / 'xception-b5690688.pth')
    torch.save(file_sd, path)
    m = X.return_pytorch04_xception(pretrained=True, weights_path=path)
    assert hasattr(m, 'last_linear') and not hasattr(m, 'fc')
    got = m.state_dict()
    for k, v in src.state_dict().items():
        kk = k.replace('fc.', 'last_linear.')
        assert got[kk].shape == v.shape and torch.equal(got[kk], v), k
    assert got['block3.rep.4.pointwise.weight'].shape == (728, 728, 1, 1)
    # the reference's default construction path
    monkeypatch.setenv('ISTVT_XCEPTION_WEIGHTS', path)
    t = TransferModel('xception')
    assert torch.equal(t.model.conv1.weight, src.conv1.weight)
    assert torch.equal(t.model.block12.rep[4].pointwise.weight, src.block12.rep[4].pointwise.weight)
    assert isinstance(t.model.last_linear, torch.nn.Sequential) and t.model.last_linear[1].out_features == 2
    monkeypatch.setenv('ISTVT_XCEPTION_WEIGHTS', str(tmp_path / 'missing.pth'))
    with pytest.raises(FileNotFoundError):
        X.return_pytorch04_xception(pretrained=True)
    with pytest.warns(UserWarning, match='no pretrained weights'):
        TransferModel('xception')                       # auto mode: default initialisation + a warning


def test_next_row_constructors_and_state_dict_names(pkg, golden_dir):
    """the whole Xception and the ablation variants keep the reference's constructor
    signatures and parameter names (the names are the keys the reference capture wrote into G7 / G8)."""
    from istvt_amd.network.vivit import module as M, vivit as V
    from istvt_amd.network import xception as X
    g7 = np.load(os.path.join(golden_dir, 'G7_xception.npz'))
    g8 = np.load(os.path.join(golden_dir, 'G8_siblings.npz'))

    def names(g, tag):
        return sorted(k[len(tag):] for k in g.files if k.startswith(tag))
    net = X.xception(pretrained=False)
    assert sorted(k for k, _ in net.named_parameters()) == names(g7, 'net.gnorm.')
    assert sorted(k for k, _ in X.Block(728, 1024, 2, 2, start_with_relu=True, grow_first=False).named_parameters()) == names(g7, 'b12.gnorm.')
    assert sorted(k for k, _ in X.Block(728, 728, 3, 1).named_parameters()) == names(g7, 'b4.gnorm.')
    dim, heads, dh = 64, 2, 32
    mods = {'attention': M.Attention(dim, heads, dh, 0.), 'temporal_only': M.TemporalOnlyAttention(dim, heads, dh, 0.),
            'transformer': V.Transformer(dim, 2, heads, dh, 2 * dim, 0.),
            'vivit': V.ViViT(19, 1, 3, 4, dim, 1, heads, 'cls', dim, dh, 0., 0., 2),
            'vanilla': V.VanillaTr(7, 1, 3, 4, dim, 1, heads, 'cls', dim, dh, 0., 0., 2)}
    for name, mod in mods.items():
        assert sorted(k for k, _ in mod.named_parameters()) == names(g8, name + '.gnorm.'), name
    assert isinstance(M.Attention(64, heads=1, dim_head=64).to_out, torch.nn.Identity)       # project_out rule, module.py:40
    with pytest.raises(RuntimeError, match='ROCm device'):
        X.Block(64, 128, 2, 2, False)(torch.zeros(1, 64, 9, 9))
    with pytest.raises(RuntimeError, match='ROCm device'):
        mods['vivit'](torch.zeros(1, 4, 64, 19, 19))


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its two ranks itself
    (child processes of `python -m torch.distributed.run`, never an exec, the parent never initialises the GPU), relays
    rank 0's ONE JSON line and the job's exit code.  --plumbing-only keeps it runnable without a GPU: process group
    (gloo), barriers, max-over-ranks timing and the line, around an empty step."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--plumbing-only'], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['plumbing_only'] is True and out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['distributed']['ranks'] == 2 and out['distributed']['backend'] == 'gloo'
    assert len(out['distributed']['per_rank_s']) == 2
    # the parent counts devices without torch / HIP (environment lists, else sysfs) and never has the
    # HIP runtime mapped when it spawns the ranks
    la = out['launcher']
    assert la['self_launched'] is True and la['parent_imported_torch'] is False and la['parent_mapped_hip_runtime'] is False
    # a failing rank must fail the parent: an unknown flag makes every rank exit 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--plumbing-only', '--no-such-flag'],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0


def test_bench_parent_counts_gpus_without_the_runtime(monkeypatch):
    """bench.visible_gpus(): *_VISIBLE_DEVICES lists (the smallest wins) capped by the KFD topology, no torch, no HIP; and
    `grep torch.cuda` finds nothing in bench.py before self_launch returns (the parent's whole code path)."""
    import importlib
    bench = importlib.import_module('bench')
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    n0, src0 = bench.visible_gpus()
    assert (n0 is None) or (isinstance(n0, int) and n0 >= 0 and 'kfd' in src0)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    n, src = bench.visible_gpus()
    assert src == 'HIP_VISIBLE_DEVICES' and n == (3 if n0 is None else min(3, n0))
    monkeypatch.setenv('ROCR_VISIBLE_DEVICES', '0')
    n, src = bench.visible_gpus()
    assert src == 'ROCR_VISIBLE_DEVICES' and n == (1 if n0 is None else min(1, n0))
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert bench.visible_gpus()[0] == 0
    src_text = open(os.path.join(ROOT, 'bench.py')).read()
    parent = src_text[:src_text.index('def pin_rank(')]            # parse(), visible_gpus(), self_launch()
    assert 'torch.cuda' not in parent and 'import torch' not in parent.replace('import torch as _t', '').replace(
        'import torch.distributed as _d', '')
    # too few devices: refused by the parent, with the source of the count (no rank is started)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'ISTVT_BENCH_REHEARSAL')}
    env['HIP_VISIBLE_DEVICES'] = '0'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'only' in (r.stderr + r.stdout) and 'GPU(s) visible' in (r.stderr + r.stdout)


def test_data_parallel_wrapper_is_refused_with_the_way_out(pkg):
    """train_CNN.py:185-186 wraps the model in nn.DataParallel for more than one device: every module of the HIP path
    refuses to be replicated (per-process caches / streams / bucket) and names torchrun as the replacement."""
    from istvt_amd.network.models import model_selection
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m = model_selection('resnet_3d', 1)
    for mod in m.modules():
        if type(mod).__module__.startswith('istvt_amd'):
            with pytest.raises(RuntimeError, match='torch.distributed.run'):
                mod._replicate_for_data_parallel()
    dp = torch.nn.DataParallel(m)             # constructing the wrapper is harmless; replication is what fails
    assert dp.module is m


def test_pad_mod_is_validated():
    """a bad ISTVT_PAD_MOD (r >= m, m <= 0, not two integers) fails at import with a ValueError instead of
    hanging the first allocation"""
    code = ("import sys; sys.path.insert(0, %r); import istvt_pkg; istvt_pkg.load(); from istvt_amd import ops; "
            "print(ops.pad_ld(728), ops.pad_ld(2912), ops.pad_ld(512))" % ROOT)
    ok = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120,
                        env=dict(os.environ, ISTVT_PAD_MOD='2,1'))
    assert ok.returncode == 0 and ok.stdout.split() == ['832', '3008', '576'], ok.stdout + ok.stderr
    for bad in ('2,2', '0,0', '-1,0', '3', 'a,b'):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, ISTVT_PAD_MOD=bad))
        assert r.returncode != 0 and 'ValueError' in r.stderr, (bad, r.stderr[-500:])


def test_dualnet_xception_surface(pkg, golden_dir):
    """network/xception_for_dualnet.py: constructor default (num_classes=1), the five split points, parameter names equal
    to the reference class's (every parameter that received a gradient in the reference capture G9 exists here with the
    same name; `fc` until get_xception() renames it to `last_linear`, :349-351)."""
    from istvt_amd.network import xception_for_dualnet as XD
    net = XD.Xception()
    assert net.num_classes == 1 and net.fc.out_features == 1 and isinstance(net.dp, torch.nn.Dropout) and net.dp.p == 0.2
    for name in ('fea_0_7', 'fea_8_12', 'fea_0_4', 'fea_5_8', 'fea_9_12', 'features', 'logits', 'forward'):
        assert callable(getattr(net, name))
    g = np.load(os.path.join(golden_dir, 'G9_dualnet_halves.npz'))
    names = dict(net.named_parameters())
    ref = [k[len('gnorm.'):] for k in g.files if k.startswith('gnorm.')]
    assert len(ref) > 150
    for k in ref:
        assert k in names, k
    assert set(names) - set(ref) == {'fc.weight', 'fc.bias'}
    m = XD.get_xception(3)
    assert 'last_linear.weight' in m.state_dict() and 'fc.weight' not in m.state_dict()
    with pytest.raises(RuntimeError, match='ROCm device'):
        net.fea_5_8(torch.zeros(1, 728, 5, 5))


def test_stale_weight_gradient_queue_is_dropped_by_a_forward_and_launched_inside_a_backward(pkg, monkeypatch):
    """a backward that aborted leaves its join entry and its QUEUED weight-gradient group behind.  Found by
    the next forward (no graph task is running) the queue must be dropped -- the gradients have been zeroed since, launching
    the old dy^T x products would corrupt the next step; found from inside another backward (a re-entrant pass) it must be
    launched."""
    from istvt_amd import functional as Fn, ops
    launched = []
    monkeypatch.setattr(ops, 'linear_wgrad_group', lambda q: launched.append(list(q)))
    monkeypatch.setitem(Fn._overlap, 'on', False)
    try:
        Fn._overlap['pending'][0] = (None, 12345)            # (main stream, task id) of a pass that never ran its callback
        Fn._overlap['queue'][0] = ['stale-entry']
        Fn._overlap['keep'][0] = ['operands']
        Fn.flush_stale_joins()                               # called from a forward: torch reports task id -1
        assert launched == [] and 0 not in Fn._overlap['pending'] and 0 not in Fn._overlap['queue'] and 0 not in Fn._overlap['keep']
        Fn._overlap['pending'][0] = (None, 12345)
        Fn._overlap['queue'][0] = ['outer-pass-entry']
        monkeypatch.setattr(Fn, '_graph_task', lambda: 777)  # inside a backward with another task id
        Fn.flush_stale_joins()
        assert launched == [['outer-pass-entry']] and 0 not in Fn._overlap['pending']
    finally:
        for k in ('pending', 'queue', 'keep'):
            Fn._overlap[k].pop(0, None)


def test_bench_accepts_the_reference_training_flags(monkeypatch):
    """SURVEY section 5 / VERDICT r3 row g: the names a train_CNN.py user types (train_CNN.py:1016-1057) on the build's
    entry point -- -mn resnet_3d -sl T -is S -bz B -opt SGD|Adam -lr -wd -d "0,1,.." -- mapped as the reference computes
    them (num_device = (len(device_no) + 1) // 2, per-device batch = batch_size // num_device, :179-181)."""
    import importlib
    bench = importlib.import_module('bench')

    def parse(*argv):
        monkeypatch.setattr(sys, 'argv', ['bench.py'] + list(argv))
        return bench.parse()
    a = parse('-mn', 'resnet_3d', '-sl', '6', '-is', '300', '-bz', '16', '-opt', 'Adam', '-lr', '0.0005', '-wd', '0.01', '-d', '0,1')
    assert (a.frames, a.size, a.batch, a.gpus, a.optimizer, a.learning_rate, a.weight_decay) == (6, 300, 8, 2, 'Adam', 0.0005, 0.01)
    a = parse('-d', '0,1,2,3,4,5,6,7', '-bz', '256')
    assert (a.gpus, a.batch) == (8, 32)                     # C3: 8 x 32 clips
    a = parse()
    assert (a.gpus, a.batch, a.frames, a.size, a.optimizer, a.learning_rate) == (1, 32, 8, 224, 'SGD', 0.001)
    with pytest.raises(SystemExit):
        parse('-mn', 'xception')
    with pytest.raises(SystemExit):
        parse('-bz', '15', '-d', '0,1')
    with pytest.raises(SystemExit):
        parse('--gpus', '4', '-d', '0,1')


def test_cu_reserve_is_a_launch_argument_not_library_state(pkg):
    """the entry points hold no global mutable state.  The CUs a persistent GEMM launch leaves to a collective
    travel in istvt_gemm's flags (bits 8..15, units of 8 CUs); the library exports no setter, ops keeps the value per
    device on the Python side and hands back the previous one for restoring (no GPU needed: the range check precedes
    every HIP call)."""
    import ctypes
    from istvt_amd import _lib, ops
    lib = _lib.lib()
    assert not hasattr(ctypes.CDLL(_lib.LIB_PATH), 'istvt_set_cu_reserve')
    assert ops.set_cu_reserve(32, 0) == 0 and ops.get_cu_reserve(0) == 32
    assert ops.set_cu_reserve(100, 0) == 32 and ops.get_cu_reserve(0) == 104        # whole XCD-multiples of 8
    assert ops.get_cu_reserve(1) == 0                                               # per device
    assert ops.set_cu_reserve(0, 0) == 104
    for bad in (-1, 193, 10000):
        with pytest.raises(RuntimeError):
            ops.set_cu_reserve(bad, 0)
    # 25 units = 200 CUs: refused by the entry point itself
    rc = lib.istvt_gemm(None, 64, 1, None, 64, 1, None, 256, 256, 256, 64, None, None, 0, None, 0, 0, 1, 1.0, None, None,
                        25 << 8, 1, None)
    assert rc == -3




def test_bench_n1_record_round_trip(tmp_path, monkeypatch):
    """the N = 1 headline record a default run leaves for a later N > 1 run on the same host (bench.n1_cache_write / _read):
    newest record of THIS host, user and code revision wins; unreadable, foreign, stale, other-revision records and symlinks
    are ignored (no fixed name in /tmp, O_NOFOLLOW, revision + configuration in the record, bounded age)"""
    import importlib
    import json as _json
    bench = importlib.import_module('bench')
    a, b = str(tmp_path / 'a' / 'n1.json'), str(tmp_path / 'b.json')
    assert not any(os.path.dirname(p) == '/tmp' for p in bench._n1_cache_paths())     # no fixed name in world-writable /tmp
    monkeypatch.setattr(bench, '_n1_cache_paths', lambda: [a, b])
    assert bench.n1_cache_read() is None
    out = {'value': 609.0, 'unit': 'clips/s', 'ms_per_step': 52.5, 'config': {'workload': 'C2: ...'},
           'cpu_baseline': {'value': 0.67, 'cores': 32, 'kind': 'port'}}
    bench.n1_cache_write(out)
    rec = bench.n1_cache_read()
    assert rec['value'] == 609.0 and rec['cpu_baseline']['value'] == 0.67 and rec['age_s'] >= 0 and rec['host'] == os.uname().nodename
    assert rec['code_rev'] == bench._code_rev() and rec['config'] == out['config'] and rec['uid'] == os.getuid()
    assert oct(os.stat(a).st_mode & 0o777) == '0o600'
    assert bench.n1_cache_read(workload='C2: ...')['value'] == 609.0 and bench.n1_cache_read(workload='C4') is None
    with open(b, 'w') as fh:                                # a record of another host: ignored
        _json.dump(dict(rec, host='elsewhere', written_unix=rec['written_unix'] + 100, value=1.0), fh)
    assert bench.n1_cache_read()['value'] == 609.0
    with open(b, 'w') as fh:                                # another code revision: ignored
        _json.dump(dict(rec, code_rev='0' * 16, written_unix=rec['written_unix'] + 100, value=2.0), fh)
    assert bench.n1_cache_read()['value'] == 609.0
    with open(b, 'w') as fh:                                # too old: ignored
        _json.dump(dict(rec, written_unix=rec['written_unix'] - bench.N1_MAX_AGE_S - 10, value=3.0), fh)
    assert bench.n1_cache_read()['value'] == 609.0
    os.remove(b)
    os.symlink(a, b)                                        # a symlink is never followed (and a is still valid)
    assert bench.n1_cache_read()['value'] == 609.0
    with open(a, 'w') as fh:
        fh.write('not json')
    assert bench.n1_cache_read() is None                    # a: unreadable, b: a symlink


def test_host_scalar_passes_host_tensors_through(pkg):
    from istvt_amd import parallel
    h = parallel.HostScalar(torch.tensor([1.5]))
    assert h.item() == 1.5 and float(h) == 1.5 and int(parallel.HostScalar(torch.tensor(3))) == 3
    with pytest.raises(ValueError):
        parallel.HostScalar(torch.zeros(3))


def test_derived_weight_layouts_follow_the_parameter(pkg):
    """ops.derived: a layout built from a parameter with a few torch ops is cached until the parameter changes (version
    counter or a raw-pointer writer's epoch), rebuilt as a FRESH tensor by the operand refresh (what a live graph holds of
    the old one stays intact), and dropped when the parameter dies."""
    import gc
    from istvt_amd import ops
    w = torch.nn.Parameter(torch.arange(12.0).reshape(3, 4))
    calls = []

    def build(q):
        calls.append(1)
        return q.detach().t().contiguous()
    key = (id(w), 'test-layout')
    a = ops.derived(key, w, build)
    assert ops.derived(key, w, build) is a and len(calls) == 1
    with torch.no_grad():
        w.mul_(2.0)                                         # an optimizer step: the version counter moves
    held = a.clone()
    ops._refresh_derived()                                  # (what refresh_stale_operands() runs first)
    b = ops.derived(key, w, build)
    assert b is not a and len(calls) == 2 and torch.equal(b, w.detach().t()) and torch.equal(a, held)
    ops.invalidate_weight_cache()                           # a kernel wrote the parameter through a raw pointer
    c = ops.derived(key, w, build)
    assert c is not b and len(calls) == 3
    del w
    gc.collect()
    assert key not in ops._derived


def test_fused_zero_grad_recovers_from_an_aborted_backward(pkg):
    """the recovery from a backward that aborted behind the grad-ready hook (orphaned early all-reduce, CU
    reserve raised) must be reachable from the DEFAULT loop, where the fused step re-zeroes and zero_grad() skips
    bucket.zero()."""
    import types
    from istvt_amd import parallel
    p = torch.nn.Parameter(torch.zeros(4))
    b = parallel.GradBucket([p], fuse_accumulate=True)

    class Work:
        waited = False

        def wait(self):
            self.waited = True

    w = Work()
    b._early_work = (w, 0)
    b.flat.fill_(3.0)
    opt = types.SimpleNamespace(bucket=b, fused_zero_grad=True, steps=5)
    parallel._FusedOptimizer.zero_grad(opt)
    assert w.waited and b._early_work is None and float(b.flat.abs().sum()) == 0.0      # dropped AND really zeroed
    b.flat.fill_(3.0)
    parallel._FusedOptimizer.zero_grad(opt)                                             # the normal loop: no zero pass
    assert float(b.flat.sum()) == 12.0


def test_operand_refresh_baseline_is_mode_independent(pkg):
    """the idle-holder baseline of ops.refresh_stale_operands() is taken under grad mode whatever mode the first
    stale refresh runs in (the optimizer step: possibly torch.inference_mode())."""
    from istvt_amd import ops
    ops._idle_counts.clear()
    with torch.inference_mode():
        a = (ops._idle_holder_counts(True), ops._idle_holder_counts(False))
    ops._idle_counts.clear()
    b = (ops._idle_holder_counts(True), ops._idle_holder_counts(False))
    assert a == b


def test_static_address_refresh_rewrites_cached_copies_in_place(pkg):
    """parallel.StepGraphs relies on ops' static-address mode: every cached copy derived from a parameter is refreshed INTO
    the buffer it lives in (captured HIP graphs hold its address).  The host logic, on CPU tensors: a plain cast, a weight_t_as
    transpose, a stacked operand, the lazily made transpose of a cast copy and a derived layout -- all rewritten in place
    after the parameters change; with the mode off, derived layouts are re-made (a live autograd graph keeps the old one)."""
    import weakref
    from istvt_amd import ops
    saved = (dict(ops._wcache), dict(ops._operands), dict(ops._derived))
    ops._wcache.clear(); ops._operands.clear(); ops._derived.clear()
    try:
        w = torch.nn.Parameter(torch.arange(12, dtype=torch.float32).reshape(3, 4))
        v = torch.nn.Parameter(torch.ones(2, 4))
        plain = w.detach().to(torch.bfloat16).clone()
        plain_t = plain.t().contiguous().clone()
        wt = w.detach().t().to(torch.bfloat16).contiguous().clone()
        cat = torch.cat((w.detach(), v.detach())).to(torch.bfloat16).clone()
        ver = ops._versions((w,))
        ops._wcache[id(w)] = (weakref.ref(w), ver, plain)
        ops._wcache[(id(plain), 'T')] = (weakref.ref(plain), 0, plain_t)
        ops._wcache[(id(w), 't')] = (weakref.ref(w), ver, wt)
        ops._wcache[((id(w), id(v)), 'cat')] = ((weakref.ref(w), weakref.ref(v)), ops._versions((w, v)), cat)
        der = ops.derived((id(w), 'tap'), w, lambda q: q.detach().t().contiguous())
        ptrs = [t.data_ptr() for t in (plain, plain_t, wt, cat, der)]
        with torch.no_grad():
            w.mul_(2.0)
            v.add_(1.0)
        # mode off: the derived layout is re-made at a new address, the plain copies are left to their lazy paths
        assert not ops.static_addresses()
        assert ops._refresh_derived() == 1 and ops._derived[(id(w), 'tap')][2].data_ptr() != ptrs[4]
        with torch.no_grad():
            w.add_(1.0)
        prev = ops.set_static_addresses(True)
        try:
            der2 = ops._derived[(id(w), 'tap')][2]
            p_der2 = der2.data_ptr()
            assert ops._refresh_derived() == 1 and ops._derived[(id(w), 'tap')][2].data_ptr() == p_der2
            assert torch.equal(der2, w.detach().t())
            assert ops._refresh_plain_copies() == 3
        finally:
            ops.set_static_addresses(prev)
        assert [t.data_ptr() for t in (plain, plain_t, wt, cat)] == ptrs[:4]
        assert torch.equal(plain, w.detach().to(torch.bfloat16)) and torch.equal(plain_t, plain.t())
        assert torch.equal(wt, w.detach().t().to(torch.bfloat16))
        assert torch.equal(cat, torch.cat((w.detach(), v.detach())).to(torch.bfloat16))
        assert ops._refresh_plain_copies() == 0                     # nothing stale any more
    finally:
        ops._wcache.clear(); ops._wcache.update(saved[0])
        ops._operands.clear(); ops._operands.update(saved[1])
        ops._derived.clear(); ops._derived.update(saved[2])
