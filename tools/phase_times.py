# Phase timing of the fused DeepFM kernels from s_memtime stamps (run with DT_AMD_STEP_STAMPS=1 on the GPU box).
import os, sys
os.environ['DT_AMD_STEP_STAMPS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from deeptables_amd.models import deepnets
from deeptables_amd._lib import lib
dev = torch.device('cuda', 0)
model = os.environ.get('MODEL', 'DeepFM')
dm = bench.build_model(getattr(deepnets, model), dev, None, bench.D, bench.MODEL_PARAMS.get(model))
batches = bench.make_batches(8192, dev, 1)
dm.model.train()
for i in range(5):       # ROWS=1: with the optimizer step (the in-step row update of the pipelined DeepFM step)
    if os.environ.get('ROWS'):
        dm.train_step([batches[i][0], batches[i][1]], batches[i][2])
    else:
        dm.forward_backward([batches[i][0], batches[i][1]], batches[i][2])
torch.cuda.synchronize()
plan = dm.fused_plan()
ws = plan._bufs[8192]['ws']
off = lib().dt_dcn_stamps_offset_floats(8192, plan.F, plan.D, plan.Nd, plan.nl) if model == 'DCN' else \
    lib().dt_deepfm_stamps_offset_floats(8192, plan.F, plan.D, plan.Nd)
tiles = 256
flat = ws[off: off + 5 * tiles * 16 * 2].cpu().numpy().view(np.uint64).astype(np.float64)
raw = list(flat[:3 * tiles * 16].reshape(3, tiles, 16)) + [flat[3 * tiles * 16:].reshape(2 * tiles, 16)]
v1 = 'DT_DEEPFM_V1' in os.environ
labels = [{0: 'entry', 1: 'staged', 2: 'gemm1', 3: 'h1 stored', 4: 'gemm2+h2', 5: 'end'},
          {0: 'entry', 1: 'prologue', 2: 'dH1', 3: 'end(dXn)'}] if v1 else \
    [{0: 'entry', 6: 'prologue loads issued', 7: 'bn params in LDS', 1: 'chunk0 staged', 2: 'gemm1 done',
      3: 'h1 in LDS', 4: 'gemm2 + partial logits (DCN: the cross forward = P GEMM + scalar steps)', 5: 'logits/loss/dz', 8: 'end (dH2, dH1, slin)', 13: 'pipelined: xhat tile + dH1 operand staged', 14: 'pipelined: dXn GEMM + partial sums done', 15: 'pipelined: dXn rows stored', 9: 'DCN: gemm2 done, cross vectors in LDS', 10: 'DCN: dH2 / dH1 done (top of the backward)', 11: 'DCN: coefficients + xhat tile in LDS', 12: 'DCN: dXc rows + G = xhat^T coeff stored'},
     ({0: 'entry (memory waves of k_wgrad_rows)', 3: 'end (rows updated / stored)'} if model == 'DeepFM' else
      {0: 'entry', 6: 'loads issued', 7: 'dH1 in LDS', 1: 'A regs', 4: 'blk0 MFMAs issued', 5: 'blk0 epilogue done (X etc. staged before it)', 8: 'all blocks done', 3: 'end (rows written)'}),
     {0: 'entry', 1: 'chunks 0,1 issued', 2: 'chunk 0 done', 3: 'K loop done', 4: 'LDS reduce done', 5: 'end (partial stored)', 6: 'pipelined: the matrix waves\' share of the row epilogue done'},
     {0: 'entry', 1: 'ids + hash insert done, row loads issued', 2: 'rows arrived, X stores issued', 3: 'row sums done', 4: 'block barrier', 5: 'end (BN partials)'}]
for k, kn in enumerate(['k_mlp_fwd', 'k_mlp_bwd'] + ([] if v1 else ['k_wgrad', 'k_sparse_fwd'])):
    st = raw[k]
    rel = st - st[:, :1]
    order = sorted([sl for sl in labels[k] if st[:, sl].max() > 0], key=lambda sl: rel[:, sl].mean())
    print(kn, 'stamps of wave 0 (shader cycles since entry; mean / min / max over the blocks, and the step from the previous stamp):')
    prev = 0.0
    for sl in order:
        m = rel[:, sl].mean()
        print(f'   {labels[k][sl]:>22s}  {m:9.0f}  {rel[:, sl].min():9.0f}  {rel[:, sl].max():9.0f}   +{m - prev:8.0f}')
        prev = m
    # s_memtime counters differ between XCDs: skew and makespan per XCD (workgroups go round-robin over the 8 XCDs)
    last = max(order, key=lambda sl: rel[:, sl].mean())
    sk, mk = [], []
    for x in range(8):
        blk = st[x::8]
        sk.append(blk[:, 0].max() - blk[:, 0].min())
        mk.append(blk[:, last].max() - blk[:, 0].min())
    print(f'   per XCD: entry skew over its blocks {np.mean(sk):.0f} (max {np.max(sk):.0f}) cycles; first entry -> last end {np.mean(mk):.0f} (max {np.max(mk):.0f}) cycles')
