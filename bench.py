#!/usr/bin/env python3
"""Throughput benchmark of the ISTVT hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]       (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     (the same ranks, started by the caller)

Workload (BASELINE.json configs[1], "C2"): per GPU B=32 clips x T=8 frames x 3x224x224, full
ISTVT (Xception stem + 12-layer decomposed spatial-temporal transformer), bf16 activation
storage / fp32 accumulate, one full training step = zero grads -> forward -> BCE loss ->
backward -> (N>1: RCCL all-reduce of the flat gradient bucket, the transformer's slice started
during the stem backward) -> SGD-momentum step (one fused launch that also re-zeroes the
gradients), on synthetic data with random-init weights, inputs resident in HBM.  Weak scaling:
the per-GPU batch is fixed, so `value` = N * B * K / max-over-ranks(time).
Other modes: --frames 16 (C4), --batch 64 --attn-fp8 (C5), --eval (forward only, a different
metric), --torch-optimizer, --no-wgrad-overlap, --no-early-allreduce (the plain variants).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (the MFMA GEMM: 97 % of
the model's FLOPs): algorithmic FLOPs of its launches / their duration measured with events on
the launch stream in an instrumented extra step.  `cpu_baseline` times the oracle (the CPU
restatement of the reference, parity-pinned to it) on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch is imported by main() AFTER the decision to self-launch: the parent of a `--gpus N` job that starts its own ranks
# never imports torch, so it cannot load -- let alone initialise -- the HIP runtime before it spawns them.
torch = None
dist = None


def _import_torch():
    global torch, dist
    if torch is None:
        import torch as _t
        import torch.distributed as _d
        torch, dist = _t, _d

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0          # HBM3E spec (MI355X_MICROARCH.md; ~6300 GB/s is the measured copy rate)
GF_PER_CLIP_FWD_BWD = {8: 1021.1, 16: 1938.7}     # SURVEY.md 8(a), T=8 / T=16 at 224^2, depth 12
GF_PER_CLIP_NATIVE = 1483.5                       # the reference's own geometry, T=6, 300^2, depth 12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--depth', type=int, default=12)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--attn-fp8', action='store_true', help='fp8 (e4m3) operands in the spatial-attention MFMAs (configs[4])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--eval', action='store_true',
                    help='SURVEY 8(f) row 1: model.eval() + no_grad forward only (train_CNN.py:837-944); a different metric')
    ap.add_argument('--no-early-allreduce', action='store_true',
                    help='N > 1: one blocking all-reduce after backward instead of starting the transformer slice early')
    ap.add_argument('--torch-optimizer', action='store_true',
                    help='torch.optim.SGD + a separate zero-grad pass instead of the fused HIP optimizer step')
    ap.add_argument('--dead-row-elimination', action='store_true',
                    help='opt-in: skip the rows of the last layer that cannot reach the logits (identical results; '
                         'STTransformer.dead_row_elimination).  Without the flag the default run still reports this '
                         'variant\'s step time as an extra field')
    ap.add_argument('--no-dre-extra', action='store_true',
                    help='skip the extra K steps that fill the with_dead_row_elimination field (profiled runs: one kind of step in the trace)')
    ap.add_argument('--no-wgrad-overlap', action='store_true',
                    help='weight-gradient GEMMs on the main stream (as in the instrumented step) instead of the side stream')
    ap.add_argument('--config', default=None, choices=['C1', 'C2', 'C3', 'C4', 'C5'],
                    help='a BASELINE.json configuration by name: C1 1-clip T=4 96^2 depth-2 float32, C2 (default) B=32 T=8 224^2 bf16, '
                         'C3 = C2 per GPU on 8 GPUs (use with --gpus 8), C4 T=16, C5 B=64 with fp8 spatial-attention operands')
    ap.add_argument('--host-boundary', action='store_true', default=True,
                    help='(default) extra field with_host_boundary: the same K steps with the reference loop\'s host side in the '
                         'timed region (train_CNN.py:506,512,534-536): per-step H2D copy of the (B,T,3,S,S) batch from pinned '
                         'memory (double-buffered on a copy stream), loss.item() and the accuracy count; NOT the headline value')
    ap.add_argument('--no-host-boundary', dest='host_boundary', action='store_false', help='skip the with_host_boundary leg')
    ap.add_argument('--keep-schedule', action='store_true', default=True,
                    help='(default since round 6) N > 1: the headline is the schedule of the FIRST timed block -- the early, '
                         'two-piece all-reduce unless --no-early-allreduce -- and the other schedule is reported as an extra')
    ap.add_argument('--auto-schedule', dest='keep_schedule', action='store_false',
                    help='N > 1: let the headline be whichever of the two all-reduce schedules measured faster (rounds 4-5 behaviour)')
    ap.add_argument('--rccl-rehearsal', action='store_true',
                    help='N = 1 only: initialise an RCCL ("nccl") process group of ONE rank and run every collective of the '
                         'N > 1 path anyway (ISTVT_FORCE_COLLECTIVES=1: parameter broadcast, the early asynchronous '
                         'all-reduce of the transformer slice from inside backward, the blocking rest, barriers, the per-rank '
                         'time gather).  On a one-GPU box this is the only way to drive RCCL itself through the calls, streams '
                         'and waits of the data-parallel step; values are unchanged and the timing says what the calls cost '
                         'without any link')
    # ---- the reference's own flag names (train_CNN.py:1016-1057), for the options that concern this path ----
    ap.add_argument('--model_name', '-mn', default='resnet_3d',
                    help="train_CNN.py -mn: 'resnet_3d' is the ISTVT model (models.py:240-282); the only one this benchmark times")
    ap.add_argument('--batch_size', '-bz', type=int, default=None,
                    help='train_CNN.py -bz: the GLOBAL batch over the -d devices (per-GPU batch = bz // devices, train_CNN.py:180-181)')
    ap.add_argument('--sequence_length', '-sl', type=int, default=None, help='train_CNN.py -sl: frames per clip (= --frames)')
    ap.add_argument('--input_size', '-is', type=int, default=None, help='train_CNN.py -is: input side (= --size)')
    ap.add_argument('--optimizer', '-opt', default='SGD', choices=['SGD', 'Adam'],
                    help='train_CNN.py -opt: SGD(momentum 0.9) or Adam (= AdamW(betas 0.9/0.999, eps 1e-8), train_CNN.py:198-201)')
    ap.add_argument('--learning_rate', '-lr', type=float, default=0.001)
    ap.add_argument('--weight_decay', '-wd', type=float, default=0.0)
    ap.add_argument('--run_device', '-d', default=None,
                    help='train_CNN.py -d "0,1,...": that many devices.  The reference wraps the model in nn.DataParallel '
                         '(train_CNN.py:185-186); here it means one process per device (= --gpus N)')
    ap.add_argument('--no-step-graphs', action='store_true',
                    help='skip the with_step_graphs leg (the same K steps with the forward / backward replayed as two captured HIP '
                         'graphs, parallel.StepGraphs, incl. the reference loop with loss.item() every step)')
    ap.add_argument('--step-graphs', action='store_true',
                    help='run the HEADLINE steps through the captured HIP graphs too (default: launch by launch; the graphs are '
                         'reported as the extra field with_step_graphs)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short C4 / C5 / C1 legs that fill the other_configs field of the default run')
    ap.add_argument('--self-launch', action='store_true',
                    help='start the rank(s) through the self-launcher even for --gpus 1 (what --gpus N > 1 does without '
                         'WORLD_SIZE in the environment): the launcher path of the multi-GPU run, testable on a one-GPU box')
    ap.add_argument('--plumbing-only', action='store_true',
                    help='exercise only the launch / process-group / timing / JSON plumbing (no model, no GPU needed): '
                         'what the CPU test of the self-launching --gpus N path runs')
    a = ap.parse_args()
    preset = {'C1': dict(batch=1, frames=4, size=96, depth=2, dtype='f32'), 'C2': {}, 'C3': {},
              'C4': dict(frames=16), 'C5': dict(batch=64, attn_fp8=True)}.get(a.config, {})
    for k, v in preset.items():
        setattr(a, k, v)
    apply_reference_flags(a)
    return a


def apply_reference_flags(a):
    """train_CNN.py's names onto this script's: -d "0,1" -> --gpus 2 (num_device = (len(device_no) + 1) // 2, :179),
    -bz B -> per-GPU batch B // devices (:181), -sl -> --frames, -is -> --size.  Anything but the ISTVT model is refused."""
    if a.model_name != 'resnet_3d':
        raise SystemExit("bench.py times the ISTVT hot path only: -mn resnet_3d (got %r; 'xception' etc. are constructible "
                         "through istvt_amd.network.models.model_selection but are not this benchmark)" % a.model_name)
    if a.run_device is not None:
        ndev = (len(a.run_device) + 1) // 2
        if a.gpus == 1:
            a.gpus = ndev
        elif a.gpus != ndev:
            raise SystemExit('bench.py: --gpus %d contradicts -d %s (%d devices)' % (a.gpus, a.run_device, ndev))
    if a.batch_size is not None:
        if a.batch_size % a.gpus:
            raise SystemExit('bench.py: -bz %d is not divisible by the %d devices' % (a.batch_size, a.gpus))
        a.batch = a.batch_size // a.gpus
    if a.sequence_length is not None:
        a.frames = a.sequence_length
    if a.input_size is not None:
        a.size = a.input_size
    return a


def visible_gpus():
    """(count, source) of the GPUs this process could use, WITHOUT touching the HIP runtime or torch: the *_VISIBLE_DEVICES
    lists if set (the smallest one wins, as the runtime applies them in turn), else the KFD topology in sysfs (nodes with
    simd_count > 0 are GPUs; CPUs have 0).  (None, reason) when neither can be read: the ranks then find out themselves."""
    counts = []
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            counts.append((len([t for t in v.split(',') if t.strip() != '']), var))
    topo = '/sys/class/kfd/kfd/topology/nodes'
    n_kfd = None
    try:
        n_kfd = 0
        for node in os.listdir(topo):
            try:
                with open(os.path.join(topo, node, 'properties')) as fh:
                    for line in fh:
                        k, _, v = line.partition(' ')
                        if k == 'simd_count':
                            n_kfd += int(v) > 0
                            break
            except (OSError, ValueError):
                pass
    except OSError:
        n_kfd = None
    if counts:
        n, var = min(counts)
        if n_kfd is not None:
            n = min(n, n_kfd)
        return n, var
    if n_kfd is not None:
        return n_kfd, 'kfd topology (sysfs)'
    return None, 'unknown (no *_VISIBLE_DEVICES, no /sys/class/kfd)'


def hip_runtime_mapped():
    """whether libamdhip64 is mapped into THIS process (it must not be in the self-launching parent)"""
    try:
        with open('/proc/self/maps') as fh:
            return any('libamdhip64' in line for line in fh)
    except OSError:
        return None


def self_launch(a):
    """`python bench.py --gpus N` (N > 1, or --self-launch) started WITHOUT torch.distributed.run: start the N ranks
    ourselves, exactly as the documented launch line does, and hand back rank 0's JSON line and the job's exit code.
    The parent never touches the GPU: it has not imported torch (see _import_torch), it counts devices from the
    environment / sysfs (visible_gpus) and it does not exec -- the ranks are child processes with inherited stdout /
    stderr.  The reference's own multi-device mode needs no launcher (nn.DataParallel, train_CNN.py:185-186); this keeps
    `--gpus N` as easy to start."""
    import socket
    import subprocess
    rehearsal = bool(os.environ.get('ISTVT_BENCH_REHEARSAL')) or a.plumbing_only
    have, source = visible_gpus()
    if not rehearsal and have is not None and have < a.gpus:
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (%s)' % (a.gpus, have, source))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this driver
    env['MASTER_ADDR'] = '127.0.0.1'
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // a.gpus)))
    env.setdefault('ISTVT_PIN_RANKS', '1')      # each rank binds itself to its own share of the cores (pin_rank below)
    # what the parent saw, for rank 0's JSON line (`launcher`)
    env['ISTVT_BENCH_LAUNCHER'] = json.dumps({'self_launched': True, 'visible_gpus': have, 'visible_gpus_source': source,
                                              'parent_imported_torch': 'torch' in sys.modules,
                                              'parent_mapped_hip_runtime': hip_runtime_mapped()})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def pin_rank(local_rank, world):
    """Bind this rank's Python (the thread that enqueues ~1400 launches per step) to its own contiguous share of the cores
    the job may use: eight ranks left to the scheduler migrate and share cores.  Returns the core list (or None)."""
    if world <= 1 or os.environ.get('ISTVT_PIN_RANKS', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


def plumbing_only(a, world, rank):
    """--plumbing-only: the rank / barrier / max-over-ranks / one-JSON-line skeleton of main() around an empty step, on
    gloo and the CPU (tests/test_host_cpu.py).  Says so in the line: it measures nothing."""
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    t0 = time.perf_counter()
    for _ in range(a.warmup + a.steps):
        if world > 1:
            dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([elapsed], dtype=torch.float64))
        per_rank = [float(t.item()) for t in allt]
    if rank == 0:
        print(json.dumps({'plumbing_only': True, 'metric': 'none (launch plumbing check)', 'value': 0.0, 'unit': 'clips/s',
                          'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                          'launcher': json.loads(os.environ.get('ISTVT_BENCH_LAUNCHER', '{"self_launched": false}')),
                          'distributed': {'backend': dist.get_backend() if world > 1 else None,
                                          'ranks': dist.get_world_size() if world > 1 else 1,
                                          'per_rank_s': [round(t, 4) for t in per_rank]}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(frames, size, depth, clips=2, reps=3):
    """oracle (kind "port"): fwd+bwd of a batch of `clips` clips of the benchmark geometry on the host cores, one
    warm-up + `reps` timed repetitions, median; bounded sample: ~1.6 s per clip on the GPU box's host ->
    (1 + 3) x 2 clips ~ 13 s."""
    from oracle import istvt_ref as R
    # torch's intra-op pool stops scaling (and then collapses) far below the 256 hardware threads of
    # the GPU box's host on this small problem: 256 threads took 209 s for the one clip, so cap at 32
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    grid = R.stem_out_side(size)
    shapes = {'xcep.model.' + k: v for k, v in R.stem_param_shapes().items()}
    shapes.update({'vit.' + k: v for k, v in R.dsttr_param_shapes(frames, grid, depth=depth).items()})
    p0 = R.random_params(shapes, seed=0)
    x = torch.randn((clips, frames, 3, size, size), generator=torch.Generator().manual_seed(0))
    labels = torch.ones(clips)
    times = []
    for rep in range(reps + 1):
        p = R.with_grad(p0)
        t0 = time.perf_counter()
        logits = R.xception_vidtr_forward(p, x, depth=depth)
        R.bce_with_logits(logits, labels).backward()
        if rep > 0:
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    # "C1 exactly as stated": 1 clip, T=4, 96x96, 2-layer ISTVT, forward only (BASELINE.json configs[0])
    g1 = R.stem_out_side(96)
    sh1 = {'xcep.model.' + k: v for k, v in R.stem_param_shapes().items()}
    sh1.update({'vit.' + k: v for k, v in R.dsttr_param_shapes(4, g1, depth=2).items()})
    p1 = R.random_params(sh1, seed=0)
    x1 = torch.randn((1, 4, 3, 96, 96), generator=torch.Generator().manual_seed(0))
    t1 = []
    with torch.no_grad():
        for rep in range(6):
            t0 = time.perf_counter()
            R.xception_vidtr_forward(p1, x1, depth=2)
            if rep > 0:
                t1.append(time.perf_counter() - t0)
    d1 = sorted(t1)[len(t1) // 2]
    c1 = {'value': round(1.0 / d1, 3), 'unit': 'clips/s', 'ms': round(d1 * 1e3, 1),
          'sample': 'C1: 1 clip, T=4, 96x96, depth 2, forward, fp32, train-mode BatchNorm; 1 warm-up + 5 repetitions, median'}
    # the reference's own geometry (T=6, 300x300, 19x19 grid, depth 12), the CPU leg of other_configs.native: 1 clip fwd+bwd
    nat = None
    if (frames, size, depth) == (8, 224, 12):
        gn = R.stem_out_side(300)
        shn = {'xcep.model.' + k: v for k, v in R.stem_param_shapes().items()}
        shn.update({'vit.' + k: v for k, v in R.dsttr_param_shapes(6, gn, depth=12).items()})
        pn0 = R.random_params(shn, seed=0)
        xn = torch.randn((1, 6, 3, 300, 300), generator=torch.Generator().manual_seed(0))
        tn = []
        for rep in range(3):
            pn = R.with_grad(pn0)
            t0 = time.perf_counter()
            R.bce_with_logits(R.xception_vidtr_forward(pn, xn, depth=12), torch.ones(1)).backward()
            if rep > 0:
                tn.append(time.perf_counter() - t0)
        dn = min(tn)
        nat = {'value': round(1.0 / dn, 4), 'unit': 'clips/s', 's_per_clip': round(dn, 2),
               'sample': 'native: 1 clip, T=6, 300x300, depth 12, fwd+bwd fp32; 1 warm-up + 2 repetitions, best'}
    host = None
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    host = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': round(clips / dt, 5), 'unit': 'clips/s', 'cores': cores, 'kind': 'port', 'c1_forward': c1, 'native': nat,
            'host': {'cpu': host, 'hardware_threads': os.cpu_count(), 'torch_threads': torch.get_num_threads()},
            'sample': 'batch of %d clips (T=%d, %dx%d, depth %d) fwd+bwd fp32, oracle/istvt_ref.py, torch %d threads: 1 warm-up + '
                      '%d repetitions, median %.1f s (all: %s)' % (clips, frames, size, size, depth, torch.get_num_threads(), reps,
                                                                 dt, ', '.join('%.1f' % t for t in times))}


def other_configs(a, local_rank, steps=10, warmup=5):
    """BASELINE.json configs the driver's one default run would otherwise never see: C4 (B=32, T=16: the long-clip
    temporal-attention stress), C5 (B=64, fp8 e4m3 operands in the spatial-attention MFMAs; the logit delta against bf16
    on identical weights and inputs) and C1 (1 clip, T=4, 96x96, depth 2, float32 forward: the reference's own
    CPU-runnable case, here on the GPU).  Each: fresh model (seed 0), `warmup` + `steps` full training steps with the
    fused SGD (C1: forwards), inputs resident in HBM.  Never the headline value."""
    import gc
    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd import ops, parallel
    from istvt_amd import stem as stem_mod
    from istvt_amd.network.vivit.vivit import XceptionVidTr
    dev = torch.device('cuda', local_rank)
    res = {}

    def free():
        ops.invalidate_weight_cache()
        gc.collect()
        torch.cuda.empty_cache()

    def leg(name, batch, frames, size, depth, dtype, fp8, forward_only):
        free()
        t_build = time.perf_counter()
        torch.manual_seed(0)
        model = XceptionVidTr(num_frames=frames, grid=stem_mod.out_side(size), depth=depth, compute_dtype=dtype,
                              attn_fp8=fp8).to(dev).train()
        g = torch.Generator(device='cpu').manual_seed(1)
        x = torch.randn((batch, frames, 3, size, size), generator=g).to(dev)
        labels = (torch.rand((batch,), generator=g) > 0.5).float().to(dev)
        rec = {'workload': '%s: B=%d T=%d %dx%d depth %d %s%s' % (name, batch, frames, size, size, depth,
                                                                   'bf16' if dtype == torch.bfloat16 else 'f32',
                                                                   ' + fp8 spatial-attention operands' if fp8 else '')}
        if fp8:
            # the same weights and inputs with and without the fp8 operands (train-mode forward, before any update)
            with torch.no_grad():
                l8 = model(x).float().view(-1)
                for m in model.modules():
                    if hasattr(m, 'attn_fp8'):
                        m.attn_fp8 = False
                l16 = model(x).float().view(-1)
                for m in model.modules():
                    if hasattr(m, 'attn_fp8'):
                        m.attn_fp8 = True
            rec['fp8_vs_bf16_logits'] = {'max_abs_delta': round(float((l8 - l16).abs().max()), 6),
                                         'max_abs_logit': round(float(l16.abs().max()), 4)}
        if forward_only:
            def step():                 # train-mode forward (BatchNorm batch statistics), as the CPU leg of cpu_baseline runs it
                with torch.no_grad():
                    return model(x).sum()
        else:
            live = [p for _, p in parallel.live_named_parameters(model)]
            bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
            opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9, weight_decay=0, zero_grad=True)
            crit = torch.nn.BCEWithLogitsLoss()

            def step():
                opt.zero_grad()
                loss = crit(model(x).view(-1), labels)
                loss.backward()
                opt.step()
                return loss
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        rec.update({'ms_per_step': round(dt * 1e3, 3), 'clips_per_s': round(batch / dt, 2), 'steps': steps, 'warmup': warmup,
                    'what': 'train-mode forward, no_grad' if forward_only else 'train step (fwd+bwd+fused SGD)',
                    'loss' if not forward_only else 'logit_sum': round(float(last.item()), 5),
                    'wall_s_incl_build': round(time.perf_counter() - t_build, 1)})
        if fp8 and not forward_only:
            # say in the line what C5 is.  The same steps with bf16 operands at the SAME batch, same box.
            for m in model.modules():
                if hasattr(m, 'attn_fp8'):
                    m.attn_fp8 = False
            for _ in range(3):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize(dev)
            dtb = (time.perf_counter() - t0) / steps
            for m in model.modules():
                if hasattr(m, 'attn_fp8'):
                    m.attn_fp8 = True
            rec['bf16_same_batch'] = {'ms_per_step': round(dtb * 1e3, 3), 'clips_per_s': round(batch / dtb, 2),
                                      'fp8_over_bf16_time': round(dt / dtb, 4)}
            rec['note'] = ('a NUMERICS configuration, not a performance one: the e4m3 operands are converted in registers and fed '
                           'to the non-scaled v_mfma_f32_16x16x32_fp8_fp8, which runs at the bf16 rate on gfx950.  The 2x form '
                           '(v_mfma_scale_f32_16x16x128_f8f6f4) contracts 128 deep per instruction: Q K^T contracts over d_head = 64, so '
                           'half of every such instruction would multiply padding (= the bf16 rate again), only P V (contraction over '
                           'keys) could use it, and both attention kernels are bound by softmax vector issue and q/k/v bytes '
                           '(0.48 / 0.39 of min(MFMA, AI x HBM)), not by the MFMA pipe; spatial attention is 2.5 ms of a 52 ms step.  '
                           'fp8 q/k/v written by the QKV GEMM would halve the attention READ bytes but the backward needs the '
                           'bf16 values (or loses gradient accuracy): not built')
        if forward_only or batch * frames <= 32:
            # launch-bound legs: the same steps replayed as captured HIP graphs (parallel.StepGraphs)
            model.enable_step_graphs(True)
            for _ in range(4):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                last_g = step()
            torch.cuda.synchronize(dev)
            dtg = (time.perf_counter() - t0) / steps
            rec['with_step_graphs'] = {'ms_per_step': round(dtg * 1e3, 3), 'clips_per_s': round(batch / dtg, 2),
                                       'same_result': bool(float(last_g.item()) == float(last.item())) if forward_only else None,
                                       'stats': dict(model._step_graphs.stats)}
            model.enable_step_graphs(False)
        gf = GF_PER_CLIP_FWD_BWD.get(frames) if (size == 224 and depth == 12 and not forward_only) else None
        if (frames, size, depth, forward_only) == (6, 300, 12, False):
            gf = GF_PER_CLIP_NATIVE
        if gf:
            rec['model_tflops'] = round(batch / dt * gf / 1e3, 1)
            rec['model_mfma_frac'] = round(batch / dt * gf / 1e3 / PEAK_BF16_TFLOPS, 4)
        res[name] = rec

    # what the reference itself runs (train_CNN.py:1049,1041 defaults -is 300 -bz 16; vivit.py:201 DSTTr(19, 1, 1, 6)): the
    # only geometry whose parity is pinned DIRECTLY by reference goldens (G5 / G6); P = 362 tokens per frame
    leg('native', 16, 6, 300, 12, torch.bfloat16, False, False)
    leg('C4', 32, 16, 224, 12, torch.bfloat16, False, False)
    leg('C5', 64, 8, 224, 12, torch.bfloat16, True, False)
    leg('C1', 1, 4, 96, 2, torch.float32, False, True)
    free()
    return res


def _n1_cache_paths():
    """per-user locations only (never a fixed name in world-writable /tmp): the repo's gpurun_out/ and $XDG_CACHE_HOME"""
    cache = os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache')
    return [os.path.join(ROOT, 'gpurun_out', 'istvt_bench_n1.json'), os.path.join(cache, 'istvt_amd', 'istvt_bench_n1.json')]


N1_MAX_AGE_S = 6 * 3600


def _code_rev():
    """what the record is tied to: sha256 of bench.py + the library (a stale record of other code must not be embedded)"""
    import hashlib
    h = hashlib.sha256()
    for path in (os.path.abspath(__file__), os.path.join(ROOT, '2023-tifs-istvt_amd', 'libistvt_hip.so')):
        try:
            with open(path, 'rb') as fh:
                h.update(fh.read())
        except OSError:
            h.update(b'missing')
    return h.hexdigest()[:16]


def n1_cache_write(out):
    """the N = 1 headline (value, ms per step, cpu_baseline) where a later N > 1 run on the same node finds it"""
    rec = {'value': out['value'], 'unit': out['unit'], 'ms_per_step': out['ms_per_step'], 'workload': out['config']['workload'],
           'cpu_baseline': out.get('cpu_baseline'), 'written_unix': int(time.time()), 'host': os.uname().nodename,
           'uid': os.getuid(), 'code_rev': _code_rev(), 'config': out['config']}
    for path in _n1_cache_paths():
        try:
            os.makedirs(os.path.dirname(path), mode=0o700, exist_ok=True)
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, 'w') as fh:
                json.dump(rec, fh)
        except OSError:
            pass


def n1_cache_read(workload=None):
    """the newest record of THIS host, user and code revision (and workload, if given) that is younger than N1_MAX_AGE_S"""
    best, rev = None, _code_rev()
    for path in _n1_cache_paths():
        try:
            fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
            with os.fdopen(fd) as fh:
                if os.fstat(fh.fileno()).st_uid != os.getuid():
                    continue
                rec = json.load(fh)
            if (rec.get('host') == os.uname().nodename and rec.get('uid') == os.getuid() and rec.get('code_rev') == rev
                    and (workload is None or rec.get('workload') == workload)
                    and 0 <= int(time.time()) - rec['written_unix'] <= N1_MAX_AGE_S
                    and (best is None or rec['written_unix'] > best['written_unix'])):
                best = rec
        except (OSError, ValueError, KeyError, TypeError):
            pass
    if best is not None:
        best['age_s'] = int(time.time()) - best['written_unix']
    return best


def config_name(a, world):
    """BASELINE.json's name for the configuration being run (C2 is the headline one)"""
    key = (a.batch, a.frames, a.size, a.depth, a.dtype, bool(a.attn_fp8))
    if key == (32, 8, 224, 12, 'bf16', False):
        return 'C3' if world == 8 else 'C2'
    if key == (32, 16, 224, 12, 'bf16', False):
        return 'C4'
    if key == (64, 8, 224, 12, 'bf16', True):
        return 'C5'
    if key == (1, 4, 96, 2, 'f32', False):
        return 'C1'
    return 'custom'


def pmc_summary_path():
    """newest committed PMC summary (profiles/pmc/rNN_pmc_whole_step_summary.json)"""
    d = os.path.join(ROOT, 'profiles', 'pmc')
    if not os.path.isdir(d):
        return None
    names = sorted(f for f in os.listdir(d) if f.endswith('_pmc_whole_step_summary.json'))
    return os.path.join(d, names[-1]) if names else None


def make_optimizer(a, parallel, bucket, live):
    """train_CNN.py:198-201: SGD(lr, momentum 0.9, weight_decay) or AdamW(lr, betas (0.9, 0.999), eps 1e-8, weight_decay)"""
    if a.torch_optimizer:
        if a.optimizer == 'Adam':
            return torch.optim.AdamW(live, lr=a.learning_rate, betas=(0.9, 0.999), eps=1e-8, weight_decay=a.weight_decay)
        return torch.optim.SGD(live, lr=a.learning_rate, momentum=0.9, weight_decay=a.weight_decay)
    if a.optimizer == 'Adam':
        return parallel.FusedAdamW(bucket, lr=a.learning_rate, betas=(0.9, 0.999), eps=1e-8, weight_decay=a.weight_decay,
                                   zero_grad=True)
    return parallel.FusedSGD(bucket, lr=a.learning_rate, momentum=0.9, weight_decay=a.weight_decay, zero_grad=True)


def headline(a, world, rank, local_rank, multi):
    """the timed run of the configuration `a` names; returns the JSON dict on rank 0 (None elsewhere)"""
    dev = torch.device('cuda', local_rank)

    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd import ops, parallel
    from istvt_amd import functional as Fn
    from istvt_amd import stem as stem_mod
    from istvt_amd.network.vivit.vivit import XceptionVidTr

    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    grid = stem_mod.out_side(a.size)
    torch.manual_seed(0)                       # identical init on every rank (+ broadcast below)
    model = XceptionVidTr(num_frames=a.frames, grid=grid, depth=a.depth, compute_dtype=dtype,
                          attn_fp8=a.attn_fp8).to(dev).train()
    parallel.broadcast_parameters(model)
    live_named = parallel.live_named_parameters(model)
    live = [p for _, p in live_named]
    bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=not a.torch_optimizer)
    opt = make_optimizer(a, parallel, bucket, live)      # fused: the update in one launch over the flat buffers, zero-grad included
    crit = torch.nn.BCEWithLogitsLoss()                                      # train_CNN.py:148
    if multi and not a.no_early_allreduce:
        # the transformer's 98.8 % of the bucket is all-reduced while the stem backward still runs (parallel.py)
        bucket.enable_early_all_reduce(next(i for i, (n, _) in enumerate(live_named) if n.startswith('vit.')))

    g = torch.Generator(device='cpu').manual_seed(1 + rank)
    x = torch.randn((a.batch, a.frames, 3, a.size, a.size), generator=g).to(dev)
    labels = (torch.rand((a.batch,), generator=g) > 0.5).float().to(dev)

    if a.no_wgrad_overlap:
        Fn.set_wgrad_overlap(False)

    if a.eval:
        model.eval()
    if a.dead_row_elimination:
        model.set_dead_row_elimination(True)
    if a.step_graphs:
        model.enable_step_graphs(True)

    def step(reduce=True, xin=None, lab=None, want_logits=False, after_loss=None):
        xin = x if xin is None else xin
        lab = labels if lab is None else lab
        if a.eval:
            with torch.no_grad():
                return model(xin).sum()
        if a.torch_optimizer:
            bucket.zero()
        else:
            opt.zero_grad()                 # a pass only before the first step: the fused step re-zeroes the gradients
        logits = model(xin)
        loss = crit(logits.view(-1), lab)
        if after_loss is not None:
            after_loss(loss, logits)
        loss.backward()
        if reduce:
            bucket.all_reduce()
        opt.step()
        return (loss, logits) if want_logits else loss

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    marks = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]       # one record per step: SURVEY 8(d)'s median
    evs[0].record()
    for i in range(a.steps):
        loss = step()
        evs[i + 1].record()
        marks.append(time.perf_counter())
    # host time to enqueue one step, taken from the first two steps after the sync: later ones include waiting for
    # room in the launch queue (the GPU is ~2.5 steps behind by then)
    t_enq = (marks[min(1, len(marks) - 1)] - t0) / min(2, len(marks))
    sync()
    elapsed = time.perf_counter() - t0
    ev_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
    ev_median = ev_ms[len(ev_ms) // 2] if len(ev_ms) % 2 else 0.5 * (ev_ms[len(ev_ms) // 2 - 1] + ev_ms[len(ev_ms) // 2])
    per_rank = None
    if multi:
        # every rank's own time for the K steps (the first N > 1 run on real hardware should diagnose itself: a slow rank,
        # a slow link or the collective show up here), then the MAX over ranks as the job's time
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank = [round(float(t.item()) / a.steps * 1e3, 3) for t in allt]
        elapsed = max(float(t.item()) for t in allt)
    loss_val = float(loss.item())

    # ---- N > 1: the same K steps under the OTHER collective schedule, so that one run decides between them: the default
    # starts the transformer's 98.8 % of the bucket from inside backward (two collectives per step: that slice, then the
    # stem's 4.4 MB), the alternative is north_star's single blocking all-reduce after backward.
    schedules = None
    if multi and not a.eval:
        def timed(k):
            sync()
            t1 = time.perf_counter()
            for _ in range(k):
                step()
            sync()
            mine_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            all_ = [torch.zeros_like(mine_) for _ in range(world)]
            dist.all_gather(all_, mine_)
            return [float(t.item()) for t in all_]
        first = 'single_blocking_allreduce' if a.no_early_allreduce else 'early_two_piece_allreduce'
        schedules = {first: {'ms_per_step': round(elapsed / a.steps * 1e3, 3), 'per_rank_ms_per_step': per_rank, 'headline': True}}
        if a.no_early_allreduce:
            bucket.enable_early_all_reduce(next(i for i, (n, _) in enumerate(live_named) if n.startswith('vit.')))
            other = 'early_two_piece_allreduce'
        else:
            bucket.disable_early_all_reduce()
            other = 'single_blocking_allreduce'
        step(); step()
        ts = timed(a.steps)
        schedules[other] = {'ms_per_step': round(max(ts) / a.steps * 1e3, 3),
                            'per_rank_ms_per_step': [round(t / a.steps * 1e3, 3) for t in ts], 'headline': False}
        # The line's `value` is the FIRST timed block's schedule (round 6, VERDICT r5 item 6: the headline must not flip
        # schedules silently, so that `collectives_per_step` of a SCALE record is what the line says); the other schedule
        # is an extra.  --auto-schedule restores "whichever ran faster" (both are the complete data-parallel step, each
        # timed over the same K steps after warm-up, max over ranks).
        if not (a.no_early_allreduce or a.keep_schedule) and max(ts) < elapsed:
            schedules[first]['headline'], schedules[other]['headline'] = False, True
            elapsed = max(ts)
            per_rank = schedules[other]['per_rank_ms_per_step']
        elif a.no_early_allreduce:
            bucket.disable_early_all_reduce()
        else:
            bucket.enable_early_all_reduce(next(i for i, (n, _) in enumerate(live_named) if n.startswith('vit.')))

    # ---- N > 1: the same K steps on every rank with NO collective (what each GPU does alone, in this very job: same
    # clocks, same host load), so that one line separates "the GPUs are slower together" from "the all-reduce costs".
    alone = None
    if multi and not a.eval:
        was_early = bucket._early is not None
        first_vit = next(i for i, (n, _) in enumerate(live_named) if n.startswith('vit.'))
        bucket.disable_early_all_reduce()
        step(reduce=False); step(reduce=False)
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step(reduce=False)
        torch.cuda.synchronize(dev)
        mine_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        all_ = [torch.zeros_like(mine_) for _ in range(world)]
        dist.all_gather(all_, mine_)
        ts = [float(t.item()) for t in all_]
        alone = {'per_rank_ms_per_step': [round(t / a.steps * 1e3, 3) for t in ts],
                 'ms_per_step': round(max(ts) / a.steps * 1e3, 3),
                 'clips_per_s_if_free': round(world * a.batch * a.steps / max(ts), 3),
                 'collective_cost_ms_per_step': round((elapsed - max(ts)) / a.steps * 1e3, 3),
                 'note': 'every rank runs the same K steps without any collective, all ranks at once; headline ms_per_step '
                         'minus this = what the all-reduce (and its overlap losses) costs per step'}
        # those steps applied rank-local gradients: put every rank back on rank 0's weights and optimizer state, so that the
        # legs that follow all-reduce gradients of ONE model again (timing does not care; the numbers should still mean something)
        if bucket.flat_params is not None:
            dist.broadcast(bucket.flat_params, 0)
            ops.invalidate_weight_cache()
        else:
            parallel.broadcast_parameters(model)
        if getattr(opt, '_state_names', None):
            for name in opt._state_names:
                dist.broadcast(getattr(opt, name), 0)
        else:                               # torch.optim.SGD / AdamW (--torch-optimizer): momentum / moments are rank-local too
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        dist.broadcast(v, 0)
        for buf in model.buffers():         # BatchNorm running statistics / counters follow rank 0 as well
            if buf.is_cuda:
                dist.broadcast(buf, 0)
        if was_early:
            bucket.enable_early_all_reduce(first_vit)

    # ---- extra field: the same K steps with the opt-in dead-row elimination (identical logits / gradients, fewer FLOPs)
    dre = None
    if not a.eval and not a.dead_row_elimination and not a.no_dre_extra and a.depth > 1:
        model.set_dead_row_elimination(True)
        step(); step()
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync()
        e2 = time.perf_counter() - t1
        if multi:
            t = torch.tensor([e2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        dre = {'ms_per_step': round(e2 / a.steps * 1e3, 3), 'clips_per_s': round(world * a.batch * a.steps / e2, 3),
               'note': 'opt-in STTransformer.dead_row_elimination: last-layer rows that cannot reach the logits skipped, '
                       'results identical; NOT the headline value'}
        model.set_dead_row_elimination(False)

    # ---- extra field: the reference loop's host boundary inside the timed region (train_CNN.py:506,512,534-536)
    hostb = None
    if a.host_boundary and not a.eval:
        shape = (a.batch, a.frames, 3, a.size, a.size)
        pin = [torch.randn(shape, generator=g).pin_memory() for _ in range(2)]
        pin_l = [(torch.rand((a.batch,), generator=g) > 0.5).float().pin_memory() for _ in range(2)]
        dbuf = [torch.empty(shape, device=dev) for _ in range(2)]
        dlab = [torch.empty((a.batch,), device=dev) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event() for _ in range(2)]

        def upload(i):                      # image.cuda() / labels.cuda() of the NEXT batch, under the current step's kernels
            with torch.cuda.stream(copy_stream):
                dbuf[i % 2].copy_(pin[i % 2], non_blocking=True)
                dlab[i % 2].copy_(pin_l[i % 2], non_blocking=True)
                ready[i % 2].record(copy_stream)

        def hb_steps(n, scoped):
            correct, running = 0, 0.0
            upload(0)
            for i in range(n):
                torch.cuda.current_stream(dev).wait_event(ready[i % 2])
                if i + 1 < n:
                    # its buffer was last read by step i - 1: complete when the .item() below has returned (plain form);
                    # in the scoped form the copy stream waits for this step's position in the launch stream instead
                    if scoped:
                        copy_stream.wait_stream(torch.cuda.current_stream(dev))
                    upload(i + 1)
                if scoped:
                    # the same numbers at the same program points, each read back behind the event of ITS producer
                    # (parallel.HostScalar) instead of behind the whole step
                    hs = {}

                    def grab(loss_t, logits_t, lab_t=dlab[i % 2], hs=hs):
                        hs['loss'] = parallel.HostScalar(loss_t)
                        hs['acc'] = parallel.HostScalar(torch.sum((logits_t.detach().view(-1) > 0).float() == lab_t))
                    step(xin=dbuf[i % 2], lab=dlab[i % 2], after_loss=grab)
                    running += float(hs['loss'])
                    correct += int(hs['acc'])
                else:
                    loss_i, logits_i = step(xin=dbuf[i % 2], lab=dlab[i % 2], want_logits=True)
                    preds = (logits_i.view(-1) > 0).float()
                    running += loss_i.item()                              # the per-step device -> host sync of the reference
                    correct += int(torch.sum(preds == dlab[i % 2]).item())
            return correct, running

        def hb_timed(scoped):
            hb_steps(2, scoped)
            sync()
            t1 = time.perf_counter()
            res = hb_steps(a.steps, scoped)
            sync()
            e3 = time.perf_counter() - t1
            if multi:
                t = torch.tensor([e3], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e3 = float(t.item())
            return e3, res
        e3, res_plain = hb_timed(False)
        e4, res_scoped = hb_timed(True)
        hostb = {'ms_per_step': round(e3 / a.steps * 1e3, 3), 'clips_per_s': round(world * a.batch * a.steps / e3, 3),
                 'h2d_MB_per_step': round(a.batch * a.frames * 3 * a.size * a.size * 4 / 1e6, 1),
                 'note': 'per step: pinned-memory H2D copy of the next batch on a copy stream (hidden under the current '
                         'step), loss.item() and the accuracy count (train_CNN.py:506,512,534-536); NOT the headline value',
                 'scoped_readback': {'ms_per_step': round(e4 / a.steps * 1e3, 3),
                                     'clips_per_s': round(world * a.batch * a.steps / e4, 3),
                                     'note': 'the same loop with the loss and the accuracy count read back through '
                                             'parallel.HostScalar (a D2H copy behind the producer\'s own event instead of '
                                             '.item() on the launch stream, which drains the whole step): same values at '
                                             'the same program points (INTEGRATION.md: a two-line change of the loop)'}}

    # ---- extra field: the same K steps with forward and backward replayed as two captured HIP graphs (parallel.StepGraphs)
    graphs = None
    if not (multi or a.eval or a.torch_optimizer or a.no_step_graphs or a.step_graphs):
        model.enable_step_graphs(True)
        for _ in range(4):                  # two launch-by-launch warm-up calls, the capture, one replay
            step()
        sync()
        t1 = time.perf_counter()
        step(); step()
        t_enq_g = (time.perf_counter() - t1) / 2
        sync()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            loss_g = step()
        sync()
        e5 = time.perf_counter() - t1
        graphs = {'ms_per_step': round(e5 / a.steps * 1e3, 3), 'clips_per_s': round(world * a.batch * a.steps / e5, 3),
                  'host_enqueue_ms_per_step': round(t_enq_g * 1e3, 3), 'loss': round(float(loss_g.item()), 5),
                  'stats': dict(model._step_graphs.stats),
                  'note': 'model.enable_step_graphs(): forward and backward replayed as two captured HIP graphs (same kernels, '
                          'same bits: tests/test_model_gpu.py::test_step_graphs_*); the caller\'s loop is unchanged; NOT the headline value'}
        if hostb is not None:
            e6, _res = hb_timed(False)
            graphs['with_host_boundary_ms_per_step'] = round(e6 / a.steps * 1e3, 3)
            graphs['with_host_boundary_note'] = ('the reference loop as written (H2D copy of the next batch, loss.item() and the '
                                                 'accuracy count every step, train_CNN.py:506,512,534-536), no HostScalar')
        model.enable_step_graphs(False)

    # ---- instrumented extra step: every GEMM launch bracketed by events on its stream
    roof = None
    kern = None
    if rank == 0 and not a.no_kernel_profile:
        ops.gemm_profile = []
        ops.kernel_profile = []
        Fn.set_wgrad_overlap(False)         # one stream: a kernel's duration is then its own, not its neighbour's
        bucket.disable_early_all_reduce()   # rank 0 alone: nothing collective may start in this step
        step(reduce=False)                  # rank 0 alone: no collective in this extra step
        torch.cuda.synchronize(dev)
        recs, ops.gemm_profile = ops.gemm_profile, None
        krecs, ops.kernel_profile = ops.kernel_profile, None
        # memory-bound and attention kernel classes: achieved GB/s (algorithmic bytes) and TFLOP/s per class
        agg = {}
        for name, e0, e1, nbytes, flops in krecs:
            d = agg.setdefault(name, [0.0, 0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += nbytes
            d[2] += flops
            d[3] += 1
        # spatial attention alone cannot reach the MFMA peak: its arithmetic intensity is P / 2 flop per byte of q, k, v, o
        # (SURVEY 7.3-2), so its roof is min(MFMA peak, AI x HBM peak) -- 788 TFLOP/s at P = 197
        p_tok = grid * grid + 1
        min_roof = min(PEAK_BF16_TFLOPS, p_tok / 2.0 * PEAK_HBM_GBPS / 1e3)
        kern = {k: {'ms_per_step': round(v[0] * 1e3, 3), 'launches': v[3], 'GBps': round(v[1] / v[0] / 1e9, 1),
                    'frac_hbm_peak': round(v[1] / v[0] / 1e9 / PEAK_HBM_GBPS, 3),
                    **({'TFLOPs': round(v[2] / v[0] / 1e12, 2),
                        'frac_mfma_peak': round(v[2] / v[0] / 1e12 / PEAK_BF16_TFLOPS, 4),      # BASELINE.json: "attn MFMA util %"
                        'min_roof_TFLOPs': round(min_roof, 1),
                        'frac_of_min_roof': round(v[2] / v[0] / 1e12 / min_roof, 4)}
                       if k.startswith('attn_spatial') else {})}
                for k, v in agg.items() if v[0] > 0}
        by = {}                                   # rocprof kernel name -> [flops, seconds, launches]
        for ev0, ev1, flops, variant, shape, kname in recs:
            d = by.setdefault(kname, [0.0, 0.0, 0])
            d[0] += flops
            d[1] += ev0.elapsed_time(ev1) * 1e-3
            d[2] += 1
        if os.environ.get('ISTVT_BENCH_SHAPES'):   # per (kernel, M, N, K) table for tuning, written beside the logs
            sh = {}
            for ev0, ev1, flops, variant, shape, kname in recs:
                d = sh.setdefault((kname, shape), [0.0, 0.0, 0])
                d[0] += flops
                d[1] += ev0.elapsed_time(ev1) * 1e-3
                d[2] += 1
            with open(os.environ['ISTVT_BENCH_SHAPES'], 'w') as fh:
                for (kname, shape), d in sorted(sh.items(), key=lambda kv: -kv[1][1]):
                    fh.write('%-44s M=%-8d N=%-5d K=%-8d x%-3d %8.3f ms  %7.1f TF/s\n'
                             % (kname, shape[0], shape[1], shape[2], d[2], d[1] * 1e3, d[0] / d[1] / 1e12))
        tot_f = sum(d[0] for d in by.values())
        tot_t = sum(d[1] for d in by.values())
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        dom = max(by, key=lambda k: by[k][1])     # the kernel with the most time per step
        f, t, n = by[dom]
        ach = f / t / 1e12
        roof = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': None, 'launches_per_step': n,
                'avg_launch_us': round(t / n * 1e6, 2), 'algorithmic_gflop_per_launch': round(f / n / 1e9, 3),
                'kernel_ms_per_step': round(t * 1e3, 3),
                'all_gemm': {'tflops': round(tot_f / tot_t / 1e12, 2), 'ms_per_step': round(tot_t * 1e3, 3),
                             'by_kernel': {k: {'tflops': round(v[0] / v[1] / 1e12, 2), 'ms': round(v[1] * 1e3, 3),
                                               'launches': v[2]} for k, v in by.items()}}}
        fam = [v for k, v in by.items() if k.startswith('gemm256q_kernel')]
        if fam:
            ff_, ft_, fn_ = sum(v[0] for v in fam), sum(v[1] for v in fam), sum(v[2] for v in fam)
            # rocprof splits ONE source kernel (the persistent NT GEMM) into a name per template instantiation; together:
            roof['family'] = {'kernel': 'gemm256q_kernel<*> (all %d instantiations of the persistent NT GEMM)' % len(fam),
                              'achieved': round(ff_ / ft_ / 1e12, 2), 'frac': round(ff_ / ft_ / 1e12 / peak, 4),
                              'ms_per_step': round(ft_ * 1e3, 3), 'launches_per_step': fn_}
        # HBM traffic per launch is not measurable from inside the process: it comes from the committed PMC
        # passes over this same command (profiles/pmc/, FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE)
        pmc = pmc_summary_path()
        is_c2 = (a.batch, a.frames, a.size, a.depth, a.dtype) == (32, 8, 224, 12, 'bf16')
        if is_c2 and pmc:
            table = json.load(open(pmc))

            def rec_of(kname):
                return table.get('void ' + kname + '(GemmArgs)') or table.get(kname + '(GemmArgs)') or table.get('void ' + kname) \
                    or table.get(kname)
            # the committed counters describe THIS code only if every GEMM kernel that just ran is in them with the same
            # launches per step; otherwise the file is stale (another kernel mix) and no traffic figure is printed
            stale = [k for k, v in by.items() if k.startswith('gemm256')
                     and (rec_of(k) is None or rec_of(k).get('FETCH_SIZE', {}).get('launches_per_step') != v[2])]
            rec = rec_of(dom)
            if not stale and rec and 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
                roof['traffic'] = round((2.0 * rec['FETCH_SIZE']['avg_KB'] + rec['WRITE_SIZE']['avg_KB']) * 1024)
                roof['traffic_unit'] = 'bytes/launch (rocprofv3 --pmc, profiles/pmc/%s)' % os.path.basename(pmc)
                if 'mfma_busy_frac' in rec:
                    roof['mfma_busy_frac'] = rec['mfma_busy_frac']
                    roof['mfma_busy_note'] = 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 4 SIMDs x 256 CUs), same PMC file'
            else:
                roof['traffic_note'] = 'profiles/pmc/%s does not match the kernels of this run (%s): not reported' % (
                    os.path.basename(pmc), ', '.join(stale) if stale else dom)

    if rank == 0:
        clips = world * a.batch * a.steps
        value = clips / elapsed
        out = {
            'metric': 'clips/sec (BxTx3x224x224 forward, eval mode)' if a.eval else 'clips/sec (BxTx3x224x224 fwd+bwd)',
            'value': round(value, 3), 'unit': 'clips/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(elapsed / a.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if dtype == torch.bfloat16 else 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: B=%d/GPU T=%d %dx%d full ISTVT (Xception stem + %d-layer DSTTr) train step '
                                   '(fwd+bwd+grad all-reduce+SGD), random-init weights'
                                   % (config_name(a, world), a.batch, a.frames, a.size, a.size, a.depth),
                       'global_batch': world * a.batch, 'frames': a.frames, 'size': a.size, 'depth': a.depth,
                       'parallelism': 'dp%d' % world, 'early_allreduce': bool(multi and not a.no_early_allreduce), 'loss': round(loss_val, 5), 'attn_fp8': bool(a.attn_fp8)},
        }
        gf = GF_PER_CLIP_FWD_BWD.get(a.frames) if (a.size == 224 and a.depth == 12) else None
        if gf and a.eval:
            gf = gf / 3.0                               # forward only (fwd+bwd = 3 x fwd)
            out['config']['workload'] = out['config']['workload'].replace('train step (fwd+bwd+grad all-reduce+SGD)',
                                                                            'eval forward (no_grad, BN running stats)')
        if gf:
            out['model_tflops_per_gpu'] = round(value / world * gf / 1e3, 2)
            out['model_mfma_frac'] = round(value / world * gf / 1e3 / (PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS), 4)
        # Python needs this long to enqueue a step's ~1500 launches; while it stays below ms_per_step the GPU is the
        # bound.  (Replaying the step as one captured HIP graph was tried: hipGraphLaunch of the 1500-node graph costs
        # the host 35 ms per replay on this ROCm, no better than the eager loop.)
        out['host_enqueue_ms_per_step'] = round(t_enq * 1e3, 3)
        # hipEvent timing, median -- one event per step on the launch stream inside the timed region (rank 0);
        # `ms_per_step` / `value` stay the contract's wall clock over exactly K steps, max over ranks
        out['ms_per_step_event_median'] = round(ev_median, 3)
        out['config']['dead_row_elimination'] = bool(a.dead_row_elimination)
        # the default model normalises every row in the final LayerNorm, as the reference does (vivit.py:100); only the
        # opt-in dead-row elimination restricts it to the rows DSTTr reads
        out['config']['final_norm_class_rows_only'] = bool(a.dead_row_elimination)
        out['config']['optimizer'] = ('AdamW' if a.optimizer == 'Adam' else 'SGD(momentum 0.9)') + (' torch' if a.torch_optimizer else ' fused')
        out['config']['lr'] = a.learning_rate
        out['config']['weight_decay'] = a.weight_decay
        if dre:
            out['with_dead_row_elimination'] = dre
        if hostb:
            out['with_host_boundary'] = hostb
        if graphs:
            out['with_step_graphs'] = graphs
        out['config']['step_graphs'] = bool(a.step_graphs)
        if multi:
            out['distributed'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                  'per_rank_ms_per_step': per_rank,
                                  'grad_bucket_MB': round(bucket.numel * 4 / 2**20, 1),
                                  'scale_folded_into_optimizer': bool(bucket.defer_scale),
                                  'rccl_rehearsal': bool(a.rccl_rehearsal),
                                  'collectives_per_step': 1 if (a.no_early_allreduce or bool(schedules and schedules.get('single_blocking_allreduce', {}).get('headline'))) else 2,
                                  'headline_rule': ('the schedule of the first timed block (the other one is the extra under `schedules`); --no-early-allreduce makes '
                                                    "north_star's single blocking all-reduce the first block, --auto-schedule reports the faster of the two") if a.keep_schedule
                                  else 'the faster of the two schedules, each timed over the same K steps (max over ranks)',
                                  'cu_reserve_during_early_allreduce': bucket.cu_reserve,
                                  'schedules': schedules, 'without_collectives': alone}
        ms = torch.cuda.memory_stats(dev)
        out['allocator'] = {'reserved_GB': round(ms.get('reserved_bytes.all.peak', 0) / 2**30, 2),
                            'device_allocs': ms.get('num_device_alloc', 0), 'device_frees': ms.get('num_device_free', 0),
                            'alloc_retries': ms.get('num_alloc_retries', 0)}
        if roof:
            out['roofline'] = roof
        if kern:
            out['kernels'] = kern
        return out
    return None


def main():
    a = parse()
    wd = float(os.environ.get('ISTVT_BENCH_WATCHDOG', '900'))
    if wd > 0:
        # a run that is still alive after this many seconds dumps every thread's Python stack to stderr (and goes on):
        # a hang in a rendezvous / collective / launcher then says where it is (the default run takes about a minute;
        # ISTVT_BENCH_WATCHDOG=0 switches it off).  It found the one hang this path has had: a tcp:// rendezvous of the
        # one-rank RCCL rehearsal under torchrun, which makes every rank a CLIENT of the agent's store.
        import faulthandler
        faulthandler.dump_traceback_later(wd, repeat=True, file=sys.stderr)
    if (a.gpus > 1 or a.self_launch) and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(a))
    _import_torch()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if a.plumbing_only:
        return plumbing_only(a, world, rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if os.environ.get('ISTVT_BENCH_REHEARSAL'):
            # functional rehearsal of the N > 1 code path on a one-GPU box: every rank on cuda:0, gloo instead of RCCL
            # (the timing of such a run means nothing)
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            # device_id: the communicator is created now, on this rank's GPU, and barrier() knows its device
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
        if a.rccl_rehearsal:
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
                s.bind(('127.0.0.1', 0))
                port = s.getsockname()[1]
            os.environ['ISTVT_FORCE_COLLECTIVES'] = '1'
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            if 'WORLD_SIZE' in os.environ and 'MASTER_PORT' in os.environ:
                # started by torch.distributed.run (--self-launch --gpus 1): the launcher's own rendezvous -- under the
                # elastic agent (TORCHELASTIC_USE_AGENT_STORE) a tcp:// init would make even rank 0 a client of a store
                # that nobody serves
                dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
            else:
                dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                        device_id=torch.device('cuda', 0))
    # `multi`: the collectives of the data-parallel step run (N > 1, or the one-rank RCCL rehearsal)
    multi = world > 1 or a.rccl_rehearsal
    pinned = pin_rank(local_rank, world)
    out = headline(a, world, rank, local_rank, multi)
    if rank == 0:
        if pinned is not None and 'distributed' in out:
            out['distributed']['cores_per_rank'] = len(pinned)
        if os.environ.get('ISTVT_BENCH_LAUNCHER'):
            out['launcher'] = json.loads(os.environ['ISTVT_BENCH_LAUNCHER'])        # what the self-launching parent saw
        elif 'WORLD_SIZE' in os.environ:
            out['launcher'] = {'self_launched': False, 'note': 'ranks started by the caller (torch.distributed.run)'}
        default_c2 = (a.batch, a.frames, a.size, a.depth, a.dtype, bool(a.attn_fp8)) == (32, 8, 224, 12, 'bf16', False)
        if world == 1 and default_c2 and not (a.eval or a.no_other_configs or a.rccl_rehearsal):
            # BASELINE.json's other single-GPU configurations, short legs after the headline (its model is freed)
            out['other_configs'] = other_configs(a, local_rank)
        if world == 1 and not a.no_cpu_baseline and not a.eval:
            out['cpu_baseline'] = cpu_baseline(a.frames, a.size, a.depth)
        if world == 1 and default_c2 and not (a.eval or a.rccl_rehearsal or a.dead_row_elimination):
            n1_cache_write(out)
        if world > 1:
            # cpu_baseline is measured at N = 1 only (bounded sample on rank 0); the N > 1 line carries the newest N = 1
            # record of this host, if a run left one, so SCALE can be cross-checked against BENCH from one line
            ref = n1_cache_read()
            out['n1_reference'] = ref or {'note': 'no N=1 run of this host, user and code revision on record; cpu_baseline was '
                                                  'measured by rank 0 of this job instead; see distributed.without_collectives '
                                                  'for the same-job single-GPU figure'}
            if (ref is None or not ref.get('cpu_baseline')) and not a.no_cpu_baseline and not a.eval:
                # a fresh 8-GPU node: the line is self-contained (the bounded sample, rank 0 only, after the timed region; the
                # other ranks wait in the closing barrier)
                out['cpu_baseline'] = cpu_baseline(a.frames, a.size, a.depth)
                out['cpu_baseline']['note'] = 'measured by rank 0 of this N > 1 job (no N = 1 record on this node)'
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
