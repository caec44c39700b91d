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

ROOT = os.path.dirname(os.path.abspath(__file__))
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


def _sysfs_card(device_index=0):
    """/sys/class/drm/cardN/device of the HIP device: matched by PCI address (a node shows every GPU of the host and their
    partitions — 64 card nodes on the round-6 boxes — while HIP sees one device)"""
    import glob
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = '%04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        want = None
    cards = sorted(glob.glob('/sys/class/drm/card[0-9]*/device/pp_dpm_sclk'))
    for c in cards:
        base = os.path.dirname(c)
        if want and os.path.basename(os.path.realpath(base)).lower().startswith(want):
            return base, want
    return (os.path.dirname(cards[0]), None) if cards else (None, want)


def gpu_clocks(base):
    """current shader / memory clock and package power from the amdgpu sysfs nodes of `base` (_sysfs_card).  One reading
    costs the driver ~1 ms (measured: tools/r6/call1.sh) — it must not sit between the warm-up and the timed region; the
    bench samples from a side thread instead (ClockSampler)."""
    import glob
    import re
    if base is None:
        return None
    out = {}

    def current(path):
        try:
            txt = open(path).read()
        except OSError:
            return None
        m = re.search(r'(\d+)\s*[Mm][Hh]z\s*\*', txt)
        return int(m.group(1)) if m else None
    out['sclk_mhz'] = current(os.path.join(base, 'pp_dpm_sclk'))
    out['mclk_mhz'] = current(os.path.join(base, 'pp_dpm_mclk'))
    for hw in glob.glob(os.path.join(base, 'hwmon', 'hwmon*')):
        for name, key, scale in (('power1_input', 'power_w', 1e-6), ('power1_average', 'power_w', 1e-6)):
            try:
                out.setdefault(key, round(int(open(os.path.join(hw, name)).read()) * scale, 1))
            except (OSError, ValueError):
                pass
    return out


class ClockSampler:
    """a side thread reading (sclk, mclk, power) of the benchmarked device from sysfs in a loop while the warm-up, the
    contract region and its repeats run — the launching thread never waits for a reading (file reads release the GIL).
    `window(t0, t1)` -> the samples whose reading finished inside [t0, t1] (perf_counter seconds)."""

    def __init__(self, device_index=0):
        import threading
        self.base, self.pci = _sysfs_card(device_index)
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.base else None

    def _run(self):
        while not self._stop.is_set():
            c = gpu_clocks(self.base)
            if c:
                self.samples.append((time.perf_counter(), c.get('sclk_mhz'), c.get('mclk_mhz'), c.get('power_w')))
            self._stop.wait(0.0005)

    def __enter__(self):
        if self._th:
            # (the launching thread must not wait 5 ms for the interpreter lock while this thread parses a reading)
            self._switch = sys.getswitchinterval()
            sys.setswitchinterval(2e-5)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join(timeout=2)
            sys.setswitchinterval(self._switch)
        return False

    def window(self, t0, t1):
        sel = [s for s in self.samples if t0 <= s[0] <= t1]

        def col(i):
            v = sorted(x[i] for x in sel if x[i] is not None)
            return None if not v else {'min': v[0], 'median': v[len(v) // 2], 'max': v[-1]}
        return {'n': len(sel), 'sclk_mhz': col(1), 'mclk_mhz': col(2), 'power_w': col(3)}


def smi_clocks():
    """the same figures from `rocm-smi` (a subprocess: ~0.3-1 s — only used OUTSIDE the timed neighbourhood, as a cross-check
    of the sysfs reader and where sysfs is not readable)"""
    import subprocess
    try:
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showperflevel', '--json'], capture_output=True,
                           text=True, timeout=20)
        j = json.loads(r.stdout)
        card = j.get('card0') or next(iter(j.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if 'sclk' in kl or 'mclk' in kl or 'fclk' in kl or 'power' in kl or 'performance' in kl:
                keep[k] = v
        return keep
    except Exception as e:          # diagnostic only
        return {'error': repr(e)[:120]}


def time_steps(loop, steps, warmup, barrier, device, repeats=0, warm_through_graph=True, sample_clocks=False):
    """EXACTLY `steps` train steps are timed after the untimed warm-up, all through the product's compiled loop
    (deeptables_amd/compiled.py: k steps per hipGraph replay, k divides `steps`).

    Round 6 — what sits between the warm-up and the timed region decides a 2 ms measurement, so that interval is now as short
    as the contract allows (barrier + synchronize) and is itself reported (`pause_before_region_us`):
      * the collector pass over the parity leg's millions of objects, the creation and first record of every timing event
        (73-80 us each on a fresh event, tools/r5/call18.sh) happen BEFORE the warm-up;
      * the warm-up runs through replays of the graph that is timed: `--warmup W` asks for W untimed steps, whole replays
        hold k, so ceil(W / k) replays run (>= W steps; `warmup_steps_run` in the line).  Round 5 ran W < k as eager steps:
        the timed region then WAS the graph's first replay ever, over k - 1 rows / segment buffers no launch had touched;
      * `repeats` more regions of `steps` steps follow the contract region back to back (sync in between, like the
        contract's brackets): `repeat_step_us` shows whether the first region is an outlier or the state of the machine;
      * shader / memory clocks and power are sampled from sysfs by a side thread while the warm-up, the region and the
        repeats run (one reading costs the driver ~1 ms: in line it would idle the GPU in front of the region).
    -> (wall seconds of the contract region, its HIP-event seconds, stats)"""
    import gc
    k = loop.k if loop.graph is not None else 1
    warm_run = warmup
    if warm_through_graph and loop.graph is not None and k > 1:
        warm_run = max(k, -(-warmup // k) * k)
    gc.collect()
    gc.disable()
    if loop.dp:
        loop.phase_events = []
    n_units = (steps // max(1, k) + 2) * (1 + repeats) + 4
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(n_units)]
    for e in pool:
        e.record()
    torch.cuda.synchronize()
    sampler = ClockSampler(device.index or 0) if sample_clocks else None
    if sampler is not None:
        sampler.__enter__()
    t_idle = time.perf_counter()
    if sampler is not None:
        time.sleep(0.004)                 # a few readings of the idle device (before anything of this leg is enqueued)
    t_w0 = time.perf_counter()
    loop.run(warm_run)
    t_w = time.perf_counter()
    barrier()
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
    evs = []

    def mark(kk):
        e = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
        e.record()                 # HIP event on the launch stream after every launch unit
        evs.append((e, kk))
    e0 = pool.pop()
    t0 = time.perf_counter()
    e0.record()
    t_e0 = time.perf_counter() - t0
    loop.run(steps, on_execution=mark)
    t_enq = time.perf_counter() - t0
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t_end = time.perf_counter()
    per, prev, units = [], e0, []
    for e, kk in evs:
        per += [prev.elapsed_time(e) * 1e3 / kk] * kk          # us per step (a replay of k steps: its mean)
        units.append(round(prev.elapsed_time(e) * 1e3, 1))
        prev = e
    gpu_s = e0.elapsed_time(evs[-1][0]) / 1e3
    # the same region again, `repeats` times, back to back
    rep_us, rep_wall_us = [], []
    for _ in range(repeats):
        ea, eb = pool.pop(), pool.pop()
        ta = time.perf_counter()
        ea.record()
        loop.run(steps)
        eb.record()
        barrier()
        torch.cuda.synchronize()
        rep_wall_us.append((time.perf_counter() - ta) * 1e6 / steps)
        rep_us.append(ea.elapsed_time(eb) * 1e3 / steps)
    t_rep = time.perf_counter()
    if sampler is not None:
        sampler.__exit__()
    gc.enable()
    per.sort()

    def pct(q):
        return per[min(len(per) - 1, int(q * len(per)))]
    several = len(evs) > 1
    stats = {'median': pct(0.5), 'p10': pct(0.1) if several else None, 'p90': pct(0.9) if several else None,
             'mean': sum(per) / len(per), 'n': len(per), 'launch_units': len(evs),
             # host clock of the timed region: all launch units enqueued after `host_enqueue_us`, the GPU's own time for them
             # `gpu_us` (HIP events), the rest of `wall_us` is launch latency before the first kernel + the final synchronize
             'host_enqueue_us': t_enq * 1e6, 'wall_us': wall * 1e6, 'gpu_us': gpu_s * 1e6, 'first_record_us': t_e0 * 1e6,
             'warmup_steps_run': warm_run,
             # host time between the last warm-up launch being enqueued and the region's first record (the warm-up's GPU
             # time + barrier + synchronize), and the part of it behind the synchronize (the GPU idles for that long)
             'pause_before_region_us': (t0 - t_w) * 1e6, 'idle_gap_before_region_us': (t0 - t_sync) * 1e6}
    if sampler is not None and sampler.base:
        stats['clocks'] = {'source': 'amdgpu sysfs pp_dpm_sclk / pp_dpm_mclk / hwmon power1_input of the HIP device (PCI ' +
                                     str(sampler.pci) + '), sampled by a side thread (~1 ms per reading)',
                           'idle': sampler.window(t_idle, t_w0), 'warmup': sampler.window(t_w0, t_sync),
                           'contract_region': sampler.window(t0, t_end),
                           'repeats': sampler.window(t_end, t_rep) if repeats else None}
    if len(units) <= 64:
        stats['unit_us'] = units                       # GPU time of each launch unit, in order (the first holds the launch latency)
    if repeats:
        srt = sorted(rep_us)
        stats['repeat_step_us'] = [round(v, 2) for v in rep_us]          # HIP-event us per step of each repeated region
        stats['repeat_wall_step_us'] = [round(v, 2) for v in rep_wall_us]
        stats['repeat_spread'] = (srt[-1] - srt[0]) / srt[len(srt) // 2]
        stats['steady_step_us'] = srt[len(srt) // 2]
    return wall, gpu_s, stats


def kernel_split(loop, n_steps=None):
    """HIP events at the launch boundaries of eagerly enqueued CHAINED steps (dt_step_trace: the library records an event in
    front of the step's first launch and behind each launch group) -> mean us of A | C | E||D | F over the traced steps.  The
    steps are real train steps on the loop's slots (step j prepares step j + 1, as inside the captured execution), enqueued
    back to back without a host synchronisation in between; they run after everything that is timed."""
    import ctypes
    from deeptables_amd._lib import lib, check
    k = loop.k
    if loop.graph is None or not loop.chained or k < 3:
        return None
    n = min(k, n_steps or k)
    loop._select(k)
    loop._gather()                     # the next k batches into the slots (advances the device cursor like a replay)
    keep = loop._slots_per_step
    loop._slots_per_step = True
    check(lib().dt_step_trace(1), 'dt_step_trace')
    try:
        for j in range(k):             # (all k: the last step of a chain prepares nothing, like the execution's)
            loop._body(j, chained=True)
        us = (ctypes.c_float * 4)()
        cnt = ctypes.c_int(0)
        check(lib().dt_step_trace_read(ctypes.cast(us, ctypes.c_void_p), 4, ctypes.cast(ctypes.pointer(cnt), ctypes.c_void_p)),
              'dt_step_trace_read')
    finally:
        lib().dt_step_trace(0)
        loop._slots_per_step = keep
    torch.cuda.synchronize()
    a, c, ed, f = (float(v) for v in us)
    merged = bool(getattr(loop, 'merged', False))
    return {'A_sparse_fwd': round(a, 2), 'C_tower': round(c, 2), 'ED_wgrad_rows': round(ed, 2),
            ('FA_finish+next_sparse_fwd' if merged else 'F_finish'): round(f, 2),
            'sum': round(a + c + ed + f, 2), 'steps_traced': int(cnt.value), 'merged_launches': merged,
            'note': 'eager chained steps, HIP events between the launches (dt_step_trace), means over the traced steps; the first '
                    'step of the chain carries the prep launch in A' +
                    (' and is the only one with a kernel A of its own: every other step\'s sparse forward runs inside the step '
                     'before it (FA)' if merged else '') + '; event records between launches cost each boundary ~1-2 us'}


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
                ((args.model == 'xDeepFM' and (os.environ.get('DT_AMD_CIN_DTYPE', '') == 'bf16' or args.cin == 'bf16')) or
                 (args.model == 'AutoInt' and args.attn == 'bf16'))
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
    abs_errs = [v.get('max_abs_logit_err') for v in out.values() if isinstance(v, dict) and 'max_abs_logit_err' in v]
    out['max_abs_logit_err_all'] = max(abs_errs) if abs_errs else None
    # north_star's bar is ABSOLUTE (logits within 1e-4 of the reference, 1e-2 in the bf16 modes); the verdict's rule scales it
    # by max(1, max |logit|) — the builder's reading for graphs whose logits sit far above 1 (DCN: +-160).  Both are stated:
    out['logits_within_absolute_bar'] = bool(abs_errs) and \
        max(abs_errs) <= (1e-2 if 'bf16' in (args.tower, args.cin, getattr(args, 'attn', None)) else 1e-4)
    out['tolerance'] = ('gather bit-exact; logits 1e-4 (north_star; 1e-2 in bf16 mode) of max(1, max |logit|): a 6-layer Cross '
                        'network puts logits far above 1 — `logits_within_absolute_bar` says whether the absolute 1e-4 (1e-2) holds '
                        'as well; gradients: ' + ' / '.join(sorted(rules)) + ' (oracle/headline.verdict); Adam 1e-3 of the step')
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
    ap.add_argument('--attn', default=None, choices=['f32', 'bf16x2', 'bf16'],
                    help="autoint_params['mfma_dtype'] of AutoInt: exact fp32 MFMA, split-bf16 (two-part operands, fp32 bars) or the "
                         "layer's bf16 mode (the line's parity bars are 1e-2 then)")
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--steps-per-graph', type=int, default=20,
                    help='train steps (consecutive batches) captured into one hipGraph replay (single process)')
    ap.add_argument('--repeats', type=int, default=5,
                    help='regions of --steps steps timed back to back BEHIND the contract region (`value` is the contract '
                         'region; `step_us.repeat_step_us` / `steady_rows_per_s` are these)')
    ap.add_argument('--legacy-warmup', action='store_true',
                    help="round 5's warm-up (A/B): warm-up steps that do not fill a replay run eagerly, so that with "
                         "--warmup < steps-per-graph the timed region is the captured graph's first replay")
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
    if args.attn is not None and args.model == 'AutoInt':
        mp = dict(MODEL_PARAMS['AutoInt'])
        mp['autoint_params'] = dict(mp['autoint_params'], mfma_dtype={'f32': 'float32'}.get(args.attn, args.attn))
        MODEL_PARAMS['AutoInt'] = mp
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
    if not args.no_graph:        # (data parallel: the loop keeps k only if the whole step — exchange included — captures)
        # k divides the TIMED steps (exactly `steps` steps are timed, all through replays); the warm-up runs through the same
        # loop — whole replays and, when k does not divide it, its last steps eagerly (untimed).  Round 4 also made k divide the
        # warm-up, which put the driver's `--steps 20 --warmup 5` on 5-step graphs: a replay's fixed cost (~10 us of idle GPU)
        # was paid twice as often as in `fit`'s default (then 10, now 20 steps per execution: tools/r5/call21.sh).  (first_replay_us: a graph's first launch
        # costs what every later one does — it is uploaded at capture time.)
        spg = max(d for d in range(1, max(1, args.steps_per_graph) + 1) if args.steps % d == 0)
    spg_req = spg
    warm_capture = 2
    # rows the loop will walk: capture warm-up + the warm-up rounded up to whole replays + the contract region + its repeats
    # + one execution of traced eager steps (kernel_split)
    total_steps = warm_capture + (args.warmup + 2 * max(spg, 1)) + args.steps * (1 + max(0, args.repeats)) + max(spg, 1)
    loop = CompiledTrainLoop(dm, feed, args.batch, spg, with_optimizer=not args.no_optimizer, use_graph=not args.no_graph,
                             graph_segments=args.graph_segments,
                             order_capacity=max(feed.n, total_steps * args.batch))
    loop.set_order(ring_order(feed, args.batch, total_steps, device))
    loop.capture(warm_steps=warm_capture)
    spg = loop.k if loop.graph is not None else 1
    wall, ev_s, step_stats = time_steps(loop, args.steps, args.warmup, barrier, device, repeats=max(0, args.repeats),
                                        warm_through_graph=not args.legacy_warmup, sample_clocks=(rank == 0))
    split = None
    if rank == 0 and world == 1 and strategy is None and not args.no_optimizer and not args.no_extras:
        try:
            split = kernel_split(loop)
        except Exception as e:          # diagnostic
            split = {'error': repr(e)[:200]}
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
            loop2 = CompiledTrainLoop(dm2, feed, args.batch, spg_req, use_graph=not args.no_graph,
                                      order_capacity=max(feed.n, (warm_capture + args.warmup + 2 * spg_req + args.steps) * args.batch))
            loop2.set_order(ring_order(feed, args.batch, warm_capture + args.warmup + 2 * spg_req + args.steps, device))
            loop2.capture(warm_steps=warm_capture)
            w2, _, st2_stats = time_steps(loop2, args.steps, args.warmup, barrier, device)
            t2 = torch.tensor([w2], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            other = {'parallelism': f'dp{world}' + ('' if sharded else '+table-rows-sharded'),
                     'rows_per_s': args.batch * args.steps * world / float(t2.item()),
                     'ms_per_step': float(t2.item()) / args.steps * 1e3, 'step_us': st2_stats, 'phases': loop2.phase_times(),
                     'dp_whole_step_graph': bool(getattr(loop2, 'dp_graph', False))}
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
                       # N > 1 structures: k whole steps (fwd + bwd, RCCL exchange, optimizer) in one hipGraph, or round 5's
                       # [graph: fwd + bwd] -> eager exchange -> [graph: optimizer]
                       'dp_whole_step_graph': bool(getattr(loop, 'dp_graph', False)) if strategy is not None else None,
                       'merged_launches': bool(getattr(loop, 'merged', False)),
                       'launches_per_step': (('3 (5 for the first step of a replay)' if getattr(loop, 'merged', False) else
                                              {True: '4 (5 for the first step of a replay)', False: '5'}[bool(getattr(loop, 'chained', False))])
                                             if type(dm.fused_plan()).__name__ in ('FusedDeepFM', 'FusedDCN') and strategy is None and
                                             not args.no_optimizer else None),
                       'timed_object': 'deeptables_amd.compiled.CompiledTrainLoop (DeepModel.fit steps_per_execution)',
                       'graph_uploaded_before_first_replay': bool(loop.uploaded),
                       'optimizer_in_timed_region': not args.no_optimizer,
                       'fused_plan': type(dm.fused_plan()).__name__ if dm.fused_plan() is not None else None,
                       'tower_mfma': {0: 'f32 (exact)', 0x80: 'bf16x3 (split-bf16 operands: six bf16 MFMAs per product forward, '
                                      'three backward — dH1, dXn and, since round 6, the weight-gradient GEMMs X^T dH1 / H1^T dH2 —, '
                                      'fp32 accumulate)', 0x200: 'bf16 (plain bf16 operands, fp32 accumulate: '
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
        if step_stats.get('steady_step_us'):
            # the repeated regions behind the contract region (same steps, same graph): what `fit` sustains
            result['steady_rows_per_s'] = args.batch * world / (step_stats['steady_step_us'] * 1e-6)
        if split is not None:
            result['kernel_split_us'] = split
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
            attn_mode = args.attn or os.environ.get('DT_AMD_AUTOINT_DTYPE') or 'bf16x2'      # (the library's default at embedding size 32)
            if args.model == 'AutoInt' and attn_mode == 'bf16x2':
                result['dtype'] = 'f32 (split-bf16 attention projections / dX / weight gradient: 16-bit operands, fp32 accumulate)'
                result['roofline']['mfma_dtype'] = 'split-bf16 (three bf16 MFMAs per product) for the projection-shaped products, f32 for scores / P V'
            if args.model == 'AutoInt' and args.attn == 'bf16':
                result['dtype'] = 'bf16 (attention layer projections / dX / weight gradient, fp32 accumulate); f32 elsewhere'
                result['roofline']['mfma_dtype'] = 'bf16 for the projection-shaped products (70 % of the flops), f32 for scores / P V'
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
                    w2, _, _ = time_steps(fb, args.steps, spg, barrier, device)
                    result['fwd_bwd_only_rows_per_s'] = args.batch * args.steps / w2
                    del fb
                if not args.no_optimizer and strategy is None and args.model in ('DeepFM', 'DCN') and args.tower is None:
                    # the same command on the two variants the headline is compared with (VERDICT r5 #1d): the unchained step
                    # (five launches: DT_AMD_CHAIN=0) and the exact-fp32 tower (all products 24-bit: `--tower f32`)
                    var = {}

                    def timed_variant(model_dm):
                        lp = CompiledTrainLoop(model_dm, feed, args.batch, spg, use_graph=not args.no_graph,
                                               order_capacity=max(feed.n, (2 + args.warmup + 2 * spg + 3 * args.steps) * args.batch))
                        lp.set_order(ring_order(feed, args.batch, 2 + args.warmup + 2 * spg + 3 * args.steps, device))
                        lp.capture(warm_steps=2)
                        wv, _, sv = time_steps(lp, args.steps, args.warmup, barrier, device, repeats=2)
                        return {'rows_per_s': args.batch * args.steps / wv, 'ms_per_step': wv / args.steps * 1e3,
                                'repeat_step_us': sv.get('repeat_step_us'), 'chained': bool(lp.chained)}
                    keep_env = os.environ.get('DT_AMD_CHAIN')
                    os.environ['DT_AMD_CHAIN'] = '0'
                    try:
                        var['unchained'] = timed_variant(dm)
                    finally:
                        if keep_env is None:
                            os.environ.pop('DT_AMD_CHAIN', None)
                        else:
                            os.environ['DT_AMD_CHAIN'] = keep_env
                    mp32 = dict(MODEL_PARAMS.get(args.model) or {})
                    mp32['dnn_params'] = {'hidden_units': ((128, 0, False), (64, 0, False)), 'activation': 'relu', 'mfma_dtype': 'f32'}
                    dm32 = build_model(nets, device, None, dim, mp32)
                    var['tower_f32'] = timed_variant(dm32)
                    var['tower_f32']['dtype'] = 'f32 (exact fp32 MFMA tower: every product 24-bit, forward and backward)'
                    del dm32
                    torch.cuda.empty_cache()
                    result['variants'] = var
            except Exception as e:   # diagnostics must not kill the contract line
                result['extras_error'] = repr(e)
        if not args.no_cpu_baseline and world == 1:
            try:
                result['cpu_baseline'] = cpu_baseline(dm, batches, args.batch)
                # like for like: the CPU leg runs forward + backward only, `value` also carries the optimizer
                if result.get('fwd_bwd_only_rows_per_s') and result['cpu_baseline'].get('value'):
                    result['cpu_baseline']['gpu_fwd_bwd_only_over_cpu'] = \
                        result['fwd_bwd_only_rows_per_s'] / result['cpu_baseline']['value']
                    result['cpu_baseline']['note'] = ('CPU leg = forward + backward (no optimizer): compare with '
                                                      'fwd_bwd_only_rows_per_s, not with `value` (fwd + bwd + Adam)')
            except Exception as e:
                result['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(result))
    barrier()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
