# -*- coding:utf-8 -*-
"""bench.py — BASELINE.json metric: training rows/s (fwd+bwd), DeepFM, Criteo-shaped synthetic
table (26 categorical x 1M vocab, 13 dense, embed_dim 16), batch 8192 per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch already resident in HBM: forward, BCE loss,
backward down to the embedding row-gradients (the IndexedSlices values), the Keras-Adam update of
every dense parameter and of the looked-up table rows and — for N > 1 — the RCCL gradient exchange.
The optimizer IS in the timed region (`config.optimizer_in_timed_region`; `--no-optimizer` times
fwd+bwd alone, and the default run also reports that as `fwd_bwd_only_rows_per_s`).  The timed object is the product's
compiled loop (deeptables_amd/compiled.py, `DeepModel.fit(steps_per_execution=k)`): k consecutive steps per captured hipGraph
over static input slots gathered from the device-resident table; `fit_rows_per_s` is DeepModel.fit itself.

Before anything is timed, rank 0 runs ONE step of the benchmarked configuration through
oracle/headline.py (CPU oracle, float64) and reports `parity` (logit / gradient / Adam errors).

Rank 0 prints ONE JSON line (see the fields at the bottom).  The `cpu_baseline` leg times the CPU
oracle (a torch-CPU op-for-op restatement of the reference graph; TensorFlow is not installable
here) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F, VOCAB, ND, D = 26, 1_000_000, 13, 16
N_BATCHES = 50


def algorithmic_bytes_per_row(n_dense_params, batch, dim=D, optimizer=True):
    """SURVEY §8(d) fwd+bwd: 4F + 4Nd + 8 + 3*4*F*D + 12*P_dense/B  (= 5,250 B/row for DeepFM at B=8192).
    With the row-sparse Adam step in the timed region SURVEY §8(d) adds F*D*4*(3 reads + 3 writes) = 9,984 B/row
    (p, m, v of the looked-up rows) -> 15,234 B/row; the dense parameters' optimizer traffic (28*P_dense/B = 220 B/row)
    is not part of that figure and is left out."""
    fwd_bwd = 4 * F + 4 * ND + 8 + 12 * F * dim + 12.0 * n_dense_params / batch
    if not optimizer:
        return fwd_bwd
    return fwd_bwd + 6 * 4 * F * dim


MFMA_PEAK_F32_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense (MI355X_MICROARCH.md)
MFMA_PEAK_BF16_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_bf16, dense


def mfma_flops_per_row(model, dim):
    """Matrix-core flops of one train step per batch row for the two MFMA-bound graphs (north_star: CIN and the AutoInt
    attention are the MFMA paths), forward + backward = 3x the forward contraction count (dgrad + wgrad):
      xDeepFM: CIN layer k contracts Z[b,d,(i,j)] W[(i,j),l]: 2 D F0 H_{k-1} H_k flops per row (layers.py:692-710)
      AutoInt: per interacting layer 4 projections 2 F D D each + scores / context 2 * 2 F F D (layers.py:119-153)"""
    if model == 'xDeepFM':
        cp = MODEL_PARAMS['xDeepFM']['cin_params']
        h, fl = F, 0
        for size in cp['cross_layer_size']:
            fl += 2 * dim * F * h * size
            # direct=False: only the first half of a layer's channels feeds the next layer (layers.py:713-718), so
            # H = [26, 64, 64] for 3 x 128 -> 49.2 MFLOP/row fwd+bwd; round 2 advanced h = size and
            # overcounted 1.83x
            h = size if cp.get('direct', False) else size // 2
        return 3.0 * fl
    if model == 'AutoInt':
        n = MODEL_PARAMS['AutoInt']['autoint_params']['num_attention']
        return 3.0 * n * (4 * 2 * F * dim * dim + 2 * 2 * F * F * dim)
    return None


# BASELINE.json configs[2..4]: the non-default layer parameters of the other benchmarked graphs
MODEL_PARAMS = {
    'xDeepFM': dict(cin_params={'cross_layer_size': (128, 128, 128), 'activation': 'relu', 'use_residual': False,
                                'use_bias': False, 'direct': False, 'reduce_D': False}),
    'AutoInt': dict(autoint_params={'num_attention': 3, 'num_heads': 4, 'dropout_rate': 0, 'use_residual': True}),
    'DCN': dict(cross_params={'num_cross_layer': 6}),
}


def build_model(nets, device, strategy=None, dim=D, extra=None):
    from deeptables_amd import functional
    from deeptables_amd.models import ModelConfig, DeepModel
    from deeptables_amd.models.metainfo import CategoricalColumn, ContinuousColumn
    functional.set_seed(20241218)
    conf = ModelConfig(nets=nets, fixed_embedding_dim=True, embeddings_output_dim=dim, embedding_dropout=0,
                       dense_dropout=0, metrics=['AUC'], distribute_strategy=strategy, **(extra or {}))
    cats = [CategoricalColumn(f'C{i}', VOCAB, dim) for i in range(F)]
    conts = [ContinuousColumn('input_continuous_all', [f'I{j}' for j in range(ND)])]
    dm = DeepModel('binary', 2, conf, cats, conts)
    dm.build(device)
    return dm


def make_batches(batch, device, seed, dist_kind='uniform'):
    """N_BATCHES distinct pre-generated batches.  'zipf': Zipf(alpha = 1.05) ranks per field through the
    inverse CDF, mapped to ids by a per-field random permutation (hot rows are scattered over the table, as with
    hashed Criteo ids) — the case the duplicate merge and the Infinity Cache matter for."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    perms = None
    if dist_kind == 'zipf':
        gp = torch.Generator(device='cpu').manual_seed(1234)          # the same id mapping on every rank
        perms = torch.stack([torch.randperm(VOCAB, generator=gp) for _ in range(F)], 1)   # [VOCAB, F]
    out = []
    for _ in range(N_BATCHES):
        if dist_kind == 'zipf':
            alpha = 1.05
            u = torch.rand(batch, F, generator=g, dtype=torch.float64)
            rank = ((u * (float(VOCAB) ** (1.0 - alpha) - 1.0) + 1.0) ** (1.0 / (1.0 - alpha)) - 1.0)
            rank = rank.clamp(0, VOCAB - 1).to(torch.int64)
            idx = torch.gather(perms, 0, rank).to(torch.int32)
        else:
            idx = torch.randint(0, VOCAB, (batch, F), generator=g, dtype=torch.int32)
        dense = torch.randn(batch, ND, generator=g)
        y = (torch.rand(batch, 1, generator=g) < 0.25).float()
        out.append((idx.to(device), dense.to(device), y.to(device)))
    return out


def make_feed(batches):
    """the pre-generated batches as ONE device-resident table (training.TableBatches, resident mode): what `DeepModel.fit`
    trains on; batch i = rows [i*B, (i+1)*B)"""
    from deeptables_amd.training import TableBatches
    idx = torch.cat([b[0] for b in batches])
    dense = torch.cat([b[1] for b in batches])
    y = torch.cat([b[2] for b in batches])
    return TableBatches.from_device([idx, dense], ['cat', 'cont'], y, y_ndim=2)


def ring_order(feed, batch, n_steps, device):
    """row order of n_steps consecutive steps over the ring of pre-generated batches (batch 0, 1, ..., N-1, 0, ...)"""
    return torch.arange(n_steps * batch, device=device, dtype=torch.int64) % feed.n


def spin_gpu(device, ms=60.0):
    """~`ms` of light streaming work that touches nothing of the model: the timed region starts with the clocks up (a
    20-step timed region is 2.5 ms long; an idle GPU spends it ramping)"""
    buf = torch.empty(16 << 20, dtype=torch.float32, device=device)       # 64 MB
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        buf.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1) / 20, 1e-3)
    for _ in range(int(ms / per) + 1):
        buf.add_(1.0)


def time_steps(loop, steps, warmup, barrier, device, spin=True):
    """EXACTLY `steps` train steps are timed after `warmup` untimed ones, all through the product's compiled loop
    (deeptables_amd/compiled.py: k steps per hipGraph replay, k divides `steps`; the warm-up runs through replays of the same
    graph and, for what k does not divide, eager steps)."""
    import gc
    loop.run(warmup)
    if spin and os.environ.get('DT_BENCH_SPIN'):      # measured (tools/r4/call13.sh, tools/r5/call23.sh): under 1 % on the first replay
        spin_gpu(device)
    barrier()
    torch.cuda.synchronize()
    # host hygiene of a 2.3 ms timed region: the parity leg in front of it leaves millions of Python objects behind — a
    # generation-2 collection inside the region cost ~300 us of host clock (enqueue 501 us against 180 us without the leg)
    gc.collect()
    gc.disable()
    if loop.dp:
        loop.phase_events = []
    evs = []
    # the timing events exist (and have been recorded once) before the region starts: the first record of a fresh HIP event
    # cost the host 73-80 us (tools/r5/call18.sh: 116 -> 112 us per step on the 20-step command, same box) — the bench's own
    # overhead, not the step's.  (Polling the last event instead of sleeping in synchronize gained nothing: same call.)
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(steps // max(1, getattr(loop, 'k', 1)) + 2)]
    for e in pool:
        e.record()
    torch.cuda.synchronize()

    def mark(k):
        e = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
        e.record()                 # HIP event on the launch stream after every launch unit (median / p10 / p90 below)
        evs.append((e, k))
    e0 = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    t_e0 = time.perf_counter() - t0
    loop.run(steps, on_execution=mark)
    t_enq = time.perf_counter() - t0
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gc.enable()
    per, prev, units = [], e0, []
    for e, k in evs:
        per += [prev.elapsed_time(e) * 1e3 / k] * k          # us per step (a replay of k steps: its mean)
        units.append(round(prev.elapsed_time(e) * 1e3, 1))
        prev = e
    per.sort()

    def pct(q):
        return per[min(len(per) - 1, int(q * len(per)))]
    stats = {'median': pct(0.5), 'p10': pct(0.1), 'p90': pct(0.9), 'mean': sum(per) / len(per), 'n': len(per),
             'launch_units': len(evs),
             # host clock of the timed region: all launch units enqueued after `host_enqueue_us`, the GPU's own time for them
             # `gpu_us` (HIP events), the rest of `wall_us` is launch latency before the first kernel + the final synchronize
             'host_enqueue_us': t_enq * 1e6, 'wall_us': wall * 1e6, 'gpu_us': e0.elapsed_time(evs[-1][0]) * 1e3, 'first_record_us': t_e0 * 1e6}
    if len(units) <= 64:
        stats['unit_us'] = units                       # GPU time of each launch unit, in order (the first holds the launch latency)
    return wall, e0.elapsed_time(evs[-1][0]) / 1e3, stats


TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'deepfm_traffic.json')


def pmc_traffic(args):
    """HBM bytes per launch (= per step) from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be
    read from inside the process).  Only reported for the configuration they were collected on AND while the kernel
    sources are the ones they were collected from (the json records __graft_entry__.source_hash()); stale -> null."""
    if args.model != 'DeepFM' or args.batch != 8192 or args.dist != 'uniform' or not os.path.exists(TRAFFIC_JSON):
        return None
    try:
        import __graft_entry__ as ge
        j = json.load(open(TRAFFIC_JSON))
        if j.get('source_hash') != ge.source_hash():
            return None
        return j['bytes_per_step_corrected']
    except Exception:
        return None


def parity_leg(args, device):
    """ONE train step of the benchmarked configuration (a fresh model with the benchmark's seed, batch 0 of the uniform
    stream and batch 0 of the zipf stream) compared with the CPU oracle: oracle/headline.py.  Checker only — it runs
    before anything is timed and on its own model instance."""
    from oracle import headline
    from deeptables_amd.models import deepnets
    nets = {'DeepFM': deepnets.DeepFM, 'DCN': deepnets.DCN, 'xDeepFM': deepnets.xDeepFM,
            'AutoInt': deepnets.AutoInt}[args.model]
    dim = 32 if args.model == 'AutoInt' else D
    global N_BATCHES
    keep, out = N_BATCHES, {}
    dm = build_model(nets, device, None, dim, MODEL_PARAMS.get(args.model))
    ok, rules = True, set()
    try:
        N_BATCHES = 1
        # the CIN / attention oracles cost tens of seconds per step in float64: one id distribution for those
        for kind in (('uniform', 'zipf') if args.model in ('DeepFM', 'DCN') else (args.dist,)):
            b = make_batches(args.batch, device, seed=1234, dist_kind=kind)[0]
            r = headline.check_train_step(dm, b)
            bf16 = 'tower' if args.tower == 'bf16' else \
                (args.model == 'xDeepFM' and (os.environ.get('DT_AMD_CIN_DTYPE', '') == 'bf16' or args.cin == 'bf16'))
            good, rule = headline.verdict(r, bf16=bf16)
            ok = ok and good
            rules.add(rule)
            out[kind] = {k: (float(f'{v:.3e}') if isinstance(v, float) else v) for k, v in r.items()}
            if dm.fused_plan() is not None:
                # what the timed region runs: the optimizer step INSIDE the fused step (rows looked up once updated where
                # their gradient is formed, dense elements / segments / state in the last launch) against the separate
                # path checked just above — same table rows, slots, dense parameters, step count (oracle/headline.py)
                b2 = make_batches(args.batch, device, seed=4321, dist_kind=kind)[0]
                ri = headline.check_rows_in_step(dm, b2)
                ri['ok'] = headline.rows_in_step_ok(ri)
                ok = ok and ri['ok']
                out[kind]['in_step_optimizer'] = {k: (float(f'{v:.3e}') if isinstance(v, float) else v) for k, v in ri.items()}
                # ... and the timed path against the ORACLE itself: two consecutive steps with the optimizer inside the step's
                # launches, rows / slots / dense parameters against keras_adam_step on the float64 oracle gradient and the
                # oracle's own running m / v (warm slots in the second step) — oracle/headline.check_in_step_vs_oracle
                if not bf16:          # (a 1e-2 gradient moves an Adam update by more than this check's 2e-3 of a step)
                    N_BATCHES = 2
                    b3 = make_batches(args.batch, device, seed=777, dist_kind=kind)
                    N_BATCHES = 1
                    rv = headline.check_in_step_vs_oracle(dm, b3)
                    ok = ok and rv['ok']
                    out[kind]['in_step_optimizer']['vs_oracle'] = {k: (float(f'{v:.3e}') if isinstance(v, float) else v)
                                                                   for k, v in rv.items()}
    finally:
        N_BATCHES = keep
    out['tolerance'] = ('gather bit-exact; logits 1e-4 (north_star; 1e-2 in bf16 mode) of max(1, max |logit|): a 6-layer Cross '
                        'network puts logits far above 1; gradients: ' + ' / '.join(sorted(rules)) + ' (oracle/headline.verdict); '
                        'Adam 1e-3 of the step')
    # what the optimizer figures above are measured against: the oracle's Adam is the row-sparse ("lazy") restatement the
    # product implements for tables beyond 4 M floats; the reference's Keras Adam densifies the IndexedSlices gradient, so
    # there every row's m / v decay at every step (DESIGN.md "Known deviations")
    out['adam_semantics'] = 'row-sparse'
    out['ok'] = bool(ok)
    del dm
    torch.cuda.empty_cache()
    return out


def kernel_breakdown(dm, batch, device, sample):
    """HIP-event timing of each hot-path kernel in isolation on the current stream (diagnostic)."""
    from deeptables_amd import ops
    emb_layer = dm.model.layers_by_name['emb_categorical_vars_all']
    table = emb_layer.tables[f'd{D}']
    offs, voc = emb_layer.row_offset_d16, emb_layer.vocab_d16
    idx, dense, y = sample
    out = {}

    def timeit(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3   # us

    with torch.no_grad():
        emb, concat, fsum, fmo, rows = ops.embed_fm_linear(idx, table, offs, voc, dense)
        t = timeit(lambda: ops.embed_fm_linear(idx, table, offs, voc, dense))
        by = batch * (4 * F + F * D * 4 + ND * 4 + F * D * 4 + (F * D + ND) * 4 + F * 4 + 4 + F * 8)
        out['embed_fm_linear_fwd'] = {'us': t, 'bytes': by, 'GBps': by / t / 1e3}
        t = timeit(lambda: ops.embedding_lookup(idx, table, offs, voc))
        by = batch * (4 * F + 2 * F * D * 4 + F * 8)
        out['embedding_gather'] = {'us': t, 'bytes': by, 'GBps': by / t / 1e3}
        t = timeit(lambda: ops.fm(emb))
        by = batch * (F * D * 4 + 4)
        out['fm_fwd'] = {'us': t, 'bytes': by, 'GBps': by / t / 1e3}
    return out


def cpu_baseline(dm, batches, batch, sample_steps=150, max_threads=32):
    """Oracle (torch CPU, all host cores) fwd+bwd on the same synthetic batches; tables are looked up
    into row tensors first so the backward produces row-gradients (IndexedSlices-like), not 1.66 GB
    dense table gradients."""
    from oracle import bridge, reference_layers as R
    # tiny-op torch graphs scale badly past a few dozen threads (256 threads: 27 s/step on the GPU box)
    torch.set_num_threads(min(os.cpu_count(), max_threads))
    w = bridge.oracle_weights(dm, dtype=torch.float32)
    tables = w['emb_categorical_vars_all']
    cfg = bridge.oracle_config(dm)
    def mark(o):
        if torch.is_tensor(o):
            if o.is_floating_point():
                o.requires_grad_(True)
        elif isinstance(o, dict):
            for v in o.values():
                mark(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                mark(v)

    for k, v in w.items():
        if k == 'emb_categorical_vars_all':
            continue
        if k == 'bn_concat_emb_dense':
            mark(v[:2])
        else:
            mark(v)

    class RowTables:   # tables[i][col] -> rows that require grad
        def __init__(self, t):
            self.t, self.leaves = t, []

        def __getitem__(self, col):
            r = self.t[col].detach().requires_grad_(True)
            self.leaves.append(r)
            return r

    def one(b):
        idx, dense, y = (t.cpu() for t in b)
        w2 = dict(w)
        w2['emb_categorical_vars_all'] = [RowTables(t) for t in tables]
        logit, _ = R.model_forward(w2, idx.float(), dense, dm.config.nets, cfg, training=True)
        R.binary_crossentropy_from_logits(logit, y).backward()

    one(batches[0])
    t0 = time.perf_counter()
    for i in range(sample_steps):
        one(batches[(i + 1) % len(batches)])
    dt = time.perf_counter() - t0
    return {'value': batch * sample_steps / dt, 'unit': 'rows/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'host_cpus': os.cpu_count(),
            'sample': f'{sample_steps} fwd+bwd steps of batch {batch} (torch-CPU oracle, {torch.get_num_threads()} threads, '
                      f'{dt:.1f}s)'}


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def launch_ranks(args, argv=None):
    """`--gpus N` is a promise about the run, not a label (the reference's multi-GPU script sets the global batch to
    n_gpus x the per-GPU batch and lets MirroredStrategy place one replica per device: deeptables/tests/models/run_dt.py:35-44):
      * N > 1 without a launcher around this process (no WORLD_SIZE): re-exec under `python -m torch.distributed.run
        --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU, and exit with its status;
      * under a launcher: WORLD_SIZE must equal N, and N devices must be visible — anything else exits non-zero instead of
        printing an N = 1 number labelled N.
    DT_BENCH_SPAWN_PROBE=1 (tests/test_host_logic.py, no GPU): every rank joins a gloo group, rank 0 prints
    {"spawned_ranks": <all-reduced count>} and the process exits before anything touches a device.
    -> the world size this process runs in"""
    probe = os.environ.get('DT_BENCH_SPAWN_PROBE') == '1'
    if 'WORLD_SIZE' not in os.environ:
        if args.gpus <= 1:
            return 1
        if not probe and torch.cuda.device_count() < args.gpus:
            sys.exit(f'bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} device(s) visible')
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + \
              list(sys.argv[1:] if argv is None else argv)
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this host driver (RCCL needs it)
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ['WORLD_SIZE'])
    if world != args.gpus:
        sys.exit(f'bench.py --gpus {args.gpus} inside a launcher with WORLD_SIZE={world}: the two must agree')
    if probe:
        import torch.distributed as dist
        dist.init_process_group('gloo')
        t = torch.ones(1)
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({'spawned_ranks': int(t.item()), 'world_size': dist.get_world_size()}))
        dist.destroy_process_group()
        sys.exit(0)
    if torch.cuda.device_count() < (world if world > 1 else 1):
        sys.exit(f'bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} device(s) visible')
    return world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=8192)
    ap.add_argument('--model', default='DeepFM', choices=['DeepFM', 'xDeepFM', 'AutoInt', 'DCN', 'AFM', 'FiBiNet', 'FGCNN', 'PNN'])
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'zipf'])
    ap.add_argument('--tower', default=None, choices=['f32', 'bf16x3', 'bf16'],
                    help="dnn_params['mfma_dtype'] of the fused DeepFM / DCN step: exact-fp32 MFMA, the split-bf16 tower "
                         "(csrc/tower_x3.h) or its plain-bf16 mode (north_star's 1e-2 mode: the line's parity bars are "
                         "1e-2 then); default: the library's")
    ap.add_argument('--cin', default=None, choices=['f32', 'bf16x3', 'bf16'],
                    help="cin_params['mfma_dtype'] of xDeepFM: exact-fp32 MFMA, split-bf16 (fp32 bars) or plain bf16 (1e-2 bars)")
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--steps-per-graph', type=int, default=20,
                    help='train steps (consecutive batches) captured into one hipGraph replay (single process)')
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd+bwd only (no Adam step)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle check of the benchmarked configuration')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--tables', default='auto', choices=['auto', 'replicated', 'sharded'],
                    help='N>1: replicated tables + dense all-reduce + deduped sparse all-gather (the reference\'s '
                         'MirroredStrategy shape) or embedding rows owned per field by one rank (all-to-all). auto = sharded '
                         'for the headline DeepFM step (the only layout whose per-rank exchange and row updates do not grow '
                         'with N: DESIGN.md §5), replicated for the other models')
    ap.add_argument('--bucket-ratio', type=float, default=1.0,
                    help='N>1, replicated tables: wire size of a rank\'s sparse bucket / its lookups (1.0 never overflows)')
    ap.add_argument('--graph-segments', action='store_true',
                    help='--tables sharded: capture the launches between the collectives into hipGraphs (default: eager)')
    ap.add_argument('--force-dp', action='store_true',
                    help='N=1: run the data-parallel step structure anyway (world size 1: collectives are no-ops)')
    ap.add_argument('--force-sharded', action='store_true', help='N=1: run the sharded-table step anyway (eager)')
    args = ap.parse_args()

    world = launch_ranks(args)
    tables_auto = args.tables == 'auto'
    if args.tables == 'auto':
        args.tables = 'sharded' if (args.model == 'DeepFM' and not args.force_dp) else 'replicated'
    strategy = None
    if world > 1 or args.force_sharded or args.force_dp:
        from deeptables_amd.parallel import DataParallelStrategy, ShardedEmbeddingStrategy
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        cls = ShardedEmbeddingStrategy if (args.tables == 'sharded' or args.force_sharded) else DataParallelStrategy
        strategy = cls.from_env('nccl')
        strategy.assume_uniform_batches = True      # fixed batch per rank: no count exchange / host sync
        strategy.sparse_bucket_ratio = args.bucket_ratio
        if args.force_sharded:
            strategy.force = True
        if args.force_dp:
            strategy.force_dp = True
        device = strategy.device
        rank = strategy.rank
    else:
        device = torch.device('cuda', 0)
        torch.cuda.set_device(device)
        rank = 0
    import torch.distributed as dist
    rccl_ranks = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if rccl_ranks != world:
        sys.exit(f'bench.py --gpus {args.gpus}: the process group holds {rccl_ranks} rank(s)')

    def barrier():
        if world > 1:
            dist.barrier()

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    barrier()

    from deeptables_amd.models import deepnets
    nets = {'DeepFM': deepnets.DeepFM, 'xDeepFM': deepnets.xDeepFM, 'AutoInt': deepnets.AutoInt,
            'DCN': deepnets.DCN, 'AFM': deepnets.AFM, 'FiBiNet': deepnets.FiBiNet, 'FGCNN': deepnets.FGCNN,
            'PNN': deepnets.PNN}[args.model]
    dim = 32 if args.model == 'AutoInt' else D
    if args.tower is not None:
        mp = dict(MODEL_PARAMS.get(args.model) or {})
        mp['dnn_params'] = {'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu', 'mfma_dtype': args.tower}
        MODEL_PARAMS[args.model] = mp
    if args.cin is not None and args.model == 'xDeepFM':
        mp = dict(MODEL_PARAMS['xDeepFM'])
        mp['cin_params'] = dict(mp['cin_params'], mfma_dtype={'f32': 'float32'}.get(args.cin, args.cin))
        MODEL_PARAMS['xDeepFM'] = mp
    parity = None
    if rank == 0 and world == 1 and not args.no_parity and args.model in ('DeepFM', 'DCN', 'xDeepFM', 'AutoInt'):
        try:
            parity = parity_leg(args, device)
        except Exception as e:      # the checker must not kill the contract line; the failure is reported in it
            parity = {'ok': False, 'error': repr(e)}
    dm = build_model(nets, device, strategy, dim, MODEL_PARAMS.get(args.model))
    if strategy is not None:
        strategy.broadcast_parameters(dm.model)
    batches = make_batches(args.batch, device, seed=1234 + rank, dist_kind=args.dist)

    sharded = getattr(strategy, 'sharded_embeddings', False) and strategy.active and dm.fused_plan() is not None
    # The timed object is the PRODUCT's compiled loop (deeptables_amd/compiled.py — what DeepModel.fit(steps_per_execution=k)
    # runs): k consecutive train steps per captured hipGraph over static input slots filled from the device-resident table.
    # k = the largest value <= --steps-per-graph dividing the timed steps, so that exactly `steps` steps are timed, all through
    # replays (row-owned tables: the step stays eager unless --graph-segments — a replay's fixed cost exceeds six eager
    # launches there).
    from deeptables_amd.compiled import CompiledTrainLoop
    feed = make_feed(batches)
    spg = 1
    if world == 1 and strategy is None and not args.no_graph:
        # k divides the TIMED steps (exactly `steps` steps are timed, all through replays); the warm-up runs through the same
        # loop — whole replays and, when k does not divide it, its last steps eagerly (untimed).  Round 4 also made k divide the
        # warm-up, which put the driver's `--steps 20 --warmup 5` on 5-step graphs: a replay's fixed cost (~10 us of idle GPU)
        # was paid twice as often as in `fit`'s default (then 10, now 20 steps per execution: tools/r5/call21.sh).  (first_replay_us: a graph's first launch
        # costs what every later one does — it is uploaded at capture time.)
        spg = max(d for d in range(1, max(1, args.steps_per_graph) + 1) if args.steps % d == 0)
    warm_capture = 2
    loop = CompiledTrainLoop(dm, feed, args.batch, spg, with_optimizer=not args.no_optimizer, use_graph=not args.no_graph,
                             graph_segments=args.graph_segments,
                             order_capacity=max(feed.n, (warm_capture + args.warmup + args.steps) * args.batch))
    loop.set_order(ring_order(feed, args.batch, warm_capture + args.warmup + args.steps, device))
    loop.capture(warm_steps=warm_capture)
    spg = loop.k if loop.graph is not None else 1
    wall, ev_s, step_stats = time_steps(loop, args.steps, args.warmup, barrier, device)
    if strategy is not None and hasattr(strategy, 'check_sparse_overflow'):
        strategy.check_sparse_overflow()            # a bucket that dropped entries invalidates the run: fail loudly
    if dm.fused_plan() is not None and hasattr(dm.fused_plan(), 'check_dedupe'):
        dm.fused_plan().check_dedupe()              # an election table that overflowed (B > 8192 only) invalidates it as well
    t = torch.tensor([wall], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    rows = args.batch * args.steps * world
    value = rows / wall

    # N > 1 with --tables auto: the OTHER table layout on the same ranks, so that one scaling run answers both — `value` is
    # the default layout (row-owned tables for DeepFM), `other_layout` the reference's MirroredStrategy shape (replicated
    # tables + dense all-reduce + bucketed sparse all-gather, north_star's split) or vice versa.  Every rank runs it.
    other = None
    if (world > 1 or (args.force_sharded and os.environ.get('DT_BENCH_BOTH_LAYOUTS'))) and tables_auto and \
            args.model == 'DeepFM' and not args.no_extras and not args.no_optimizer:
        try:
            from deeptables_amd.parallel import DataParallelStrategy, ShardedEmbeddingStrategy
            cls2 = DataParallelStrategy if sharded else ShardedEmbeddingStrategy
            st2 = cls2(device=strategy.device)
            st2.assume_uniform_batches = True
            st2.sparse_bucket_ratio = args.bucket_ratio
            if world == 1:
                st2.force_dp = True
            dm2 = build_model(nets, device, st2, dim, MODEL_PARAMS.get(args.model))
            st2.broadcast_parameters(dm2.model)
            loop2 = CompiledTrainLoop(dm2, feed, args.batch, 1, use_graph=not args.no_graph,
                                      order_capacity=max(feed.n, (warm_capture + args.warmup + args.steps) * args.batch))
            loop2.set_order(ring_order(feed, args.batch, warm_capture + args.warmup + args.steps, device))
            loop2.capture(warm_steps=warm_capture)
            w2, _, st2_stats = time_steps(loop2, args.steps, args.warmup, barrier, device, spin=False)
            t2 = torch.tensor([w2], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            other = {'parallelism': f'dp{world}' + ('' if sharded else '+table-rows-sharded'),
                     'rows_per_s': args.batch * args.steps * world / float(t2.item()),
                     'ms_per_step': float(t2.item()) / args.steps * 1e3, 'step_us': st2_stats, 'phases': loop2.phase_times()}
            del loop2, dm2
            torch.cuda.empty_cache()
        except Exception as e:          # the second layout is a diagnostic: it must not kill the contract line
            other = {'error': repr(e)}

    if rank == 0:
        n_dense = sum(p.numel() for n, p in dm.model.named_parameters() if 'tables' not in n)
        bpr = algorithmic_bytes_per_row(n_dense, args.batch, dim, optimizer=not args.no_optimizer)
        step_s = ev_s / args.steps
        achieved = args.batch * bpr / step_s / 1e9
        result = {
            'metric': 'training rows/sec (fwd+bwd) DeepFM Criteo-shape batch 8192'
            if (args.model == 'DeepFM' and args.batch == 8192)
            else f'training rows/sec (fwd+bwd) {args.model} Criteo-shape batch {args.batch}',
            'value': value, 'unit': 'rows/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.model} train step = fwd+bwd' + ('' if args.no_optimizer else '+Adam') +
                                   f', Criteo-shaped synthetic: {F} cat x {VOCAB} vocab, '
                                   f'{ND} dense, embed_dim {dim}, batch {args.batch}/GPU, ids {args.dist}',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}' + ('+table-rows-sharded' if sharded else ''),
                       'rccl_ranks': rccl_ranks,
                       'hipgraph': loop.graph is not None, 'steps_per_graph_replay': spg,
                       'warmup_steps_eager': (args.warmup % spg) if loop.graph is not None else args.warmup,
                       # chained steps (deeptables_amd/compiled.py): step i of a replay runs step i + 1's election and weight layouts
                       # inside its own launches — four launches per step from the replay's second step on, five for its first
                       'chained_steps': bool(getattr(loop, 'chained', False)),
                       'launches_per_step': ({True: '4 (5 for the first step of a replay)', False: '5'}[bool(getattr(loop, 'chained', False))]
                                             if type(dm.fused_plan()).__name__ in ('FusedDeepFM', 'FusedDCN') and strategy is None and
                                             not args.no_optimizer else None),
                       'timed_object': 'deeptables_amd.compiled.CompiledTrainLoop (DeepModel.fit steps_per_execution)',
                       'graph_uploaded_before_first_replay': bool(loop.uploaded),
                       'optimizer_in_timed_region': not args.no_optimizer,
                       'fused_plan': type(dm.fused_plan()).__name__ if dm.fused_plan() is not None else None,
                       'tower_mfma': {0: 'f32 (exact)', 0x80: 'bf16x3 (split-bf16 operands: six bf16 MFMAs per product forward, '
                                      'three backward, fp32 accumulate)', 0x200: 'bf16 (plain bf16 operands, fp32 accumulate: '
                                      "north_star's 1e-2 mode)"}.get(getattr(dm.fused_plan(), 'tower_flag', 0))},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         # the PMC passes were taken on the single-process six-launch step: no figure for the N > 1 step structures
                         'traffic': pmc_traffic(args) if strategy is None else None,
                         'launch': f'one train step (fwd+bwd' + ('' if args.no_optimizer else '+Adam') +
                                   f') = one hipGraph replay / {spg} steps per replay: launch_us = HIP-event time of the timed '
                                   f'replays / the steps they hold',
                         'algorithmic_bytes_per_row': bpr, 'launch_us': step_s * 1e6},
            'step_us': step_stats,
        }
        # the arithmetic the timed path computes in: "f32" alone would hide the split-bf16 operand formats
        tflag = getattr(dm.fused_plan(), 'tower_flag', 0)
        if tflag == 0x80:
            result['dtype'] = 'f32 (split-bf16: 24-bit fwd / 16-bit bwd)'
        elif tflag == 0x200:
            result['dtype'] = 'bf16 (tower GEMM operands, fp32 accumulate); f32 elsewhere'
        fpr = mfma_flops_per_row(args.model, dim)
        if fpr is not None:      # CIN / attention graphs: the matrix cores bound the step, not HBM
            cin_mode = (MODEL_PARAMS.get('xDeepFM', {}).get('cin_params', {}).get('mfma_dtype') or
                        os.environ.get('DT_AMD_CIN_DTYPE', 'bf16x3')) if args.model == 'xDeepFM' else 'float32'
            bf16 = cin_mode == 'bf16'
            x3 = cin_mode == 'bf16x3'
            # split-bf16: the useful flops are still the fp32 contraction's; priced against the bf16 pipe they run on
            peak = MFMA_PEAK_BF16_TFLOPS if (bf16 or x3) else MFMA_PEAK_F32_TFLOPS
            tf = args.batch * fpr / step_s / 1e12
            result['roofline_hbm'] = result['roofline']
            result['roofline'] = {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
                                  'traffic': None,
                                  'mfma_dtype': ('bf16 (fp32 accumulate)' if bf16 else
                                                 'split-bf16: 6 bf16 MFMAs per product forward, 3 backward (fp32 accumulate, fp32 '
                                                 'bars); frac = useful flops / bf16 peak, issued MFMA flops are 4x that' if x3
                                                 else 'f32'),
                                  'frac_of_f32_mfma_peak': tf / MFMA_PEAK_F32_TFLOPS,
                                  'flops_per_row': fpr, 'launch_us': step_s * 1e6,
                                  'launch': f'one train step = one hipGraph replay / {spg}; flops = the CIN / attention '
                                            'contractions, fwd + dgrad + wgrad'}
            if bf16:
                result['dtype'] = 'bf16 (CIN contractions, fp32 accumulate); f32 elsewhere'
            elif x3:
                result['dtype'] = 'f32 (split-bf16 CIN: 24-bit fwd / 16-bit bwd)'

        result['first_replay_us'] = loop.first_replay_us()
        if other is not None:
            result['other_layout'] = other
            if 'rows_per_s' in other:
                result['replicated_rows_per_s' if sharded else 'sharded_rows_per_s'] = other['rows_per_s']
        ph = loop.phase_times()
        if ph is not None:          # N > 1 (or --force-dp): where a data-parallel step spends its time, on rank 0
            result['phases'] = ph
            st_ = strategy
            bucketed = getattr(st_, 'compacts_segments', False) and not sharded and os.environ.get('DT_AMD_DP_DEDUPE', '1') != '0'
            result['config']['sparse_exchange'] = (
                'per-lookup (rows, values)' if not bucketed else
                'one entry per distinct row of the rank (segments summed in place), holes skipped by the receivers'
                if st_.sparse_bucket_ratio >= 1.0 else
                'unique (row, summed grad) entries packed into a bucket of %.2f x the lookups' % st_.sparse_bucket_ratio)

        if parity is not None:
            result['parity'] = parity
        if not args.no_extras and world == 1:
            try:
                result['kernels'] = kernel_breakdown(dm, args.batch, device, batches[1]) if args.model == 'DeepFM' else {}
                if not args.no_optimizer and strategy is None:
                    # DeepModel.fit itself on the resident feed, metrics off (what a DeepTable.fit user gets): the first call
                    # captures, the timed call reuses the captured loop; shuffled epochs, loss read back once per epoch
                    keep_config, dm.config = dm.config, dm.config._replace(metrics=[])
                    try:
                        dm.fit(feed, batch_size=args.batch, epochs=1, verbose=0, shuffle=True, steps_per_execution=spg)
                        torch.cuda.synchronize()
                        ep = max(2, min(20, args.steps // 10))
                        t0 = time.perf_counter()
                        dm.fit(feed, batch_size=args.batch, epochs=ep, verbose=0, shuffle=True, steps_per_execution=spg)
                        torch.cuda.synchronize()
                        dt_fit = time.perf_counter() - t0
                        result['fit_rows_per_s'] = ep * (feed.n // args.batch) * args.batch / dt_fit
                        result['fit_note'] = (f'DeepModel.fit(feed, epochs={ep}, steps_per_execution={spg}, shuffle=True), '
                                              f'{feed.n // args.batch} steps per epoch, metrics off, wall clock incl. the '
                                              f'per-epoch permutation and loss read-back')
                    finally:
                        dm.config = keep_config
                if not args.no_optimizer:      # the same step without the Adam launches, for comparison
                    fb = CompiledTrainLoop(dm, feed, args.batch, spg, with_optimizer=False, use_graph=not args.no_graph,
                                           order_capacity=max(feed.n, (2 + spg + args.steps) * args.batch))
                    fb.set_order(ring_order(feed, args.batch, 2 + spg + args.steps, device))
                    fb.capture(warm_steps=2)
                    w2, _, _ = time_steps(fb, args.steps, spg, barrier, device, spin=False)
                    result['fwd_bwd_only_rows_per_s'] = args.batch * args.steps / w2
            except Exception as e:   # diagnostics must not kill the contract line
                result['extras_error'] = repr(e)
        if not args.no_cpu_baseline and world == 1:
            try:
                result['cpu_baseline'] = cpu_baseline(dm, batches, args.batch)
            except Exception as e:
                result['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(result))
    barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
