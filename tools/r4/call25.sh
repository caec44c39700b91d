#!/bin/bash
# round 4, call 25: the loss block of k_tower_x3 on the hardware transcendentals (the logits phase is 6.6 K of the tile's 50 K cycles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c25
O=gpurun_out/r4c25
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_fused_gpu.py tests/test_headline_gpu.py -m gpu -q -x -k "not cin and not xdeepfm and not autoint and not f32" > $O/pytest.log 2>&1
grep -E "passed|failed|error|FAILED" $O/pytest.log | tail -4
python bench.py --no-cpu-baseline > $O/line_deepfm.json 2> $O/line_deepfm.err
python bench.py --model DCN --no-cpu-baseline --no-parity > $O/line_dcn.json 2> $O/line_dcn.err
for f in deepfm dcn; do grep "^{" $O/line_$f.json | python -c "
import sys,json
j=json.loads(sys.stdin.read()); p=(j.get('parity') or {}).get('uniform',{}); print('$f', round(j['value']/1e6,3), 'M rows/s', round(j['ms_per_step']*1e3,1), 'us', 'median', round(j['step_us']['median'],1), 'parity', (j.get('parity') or {}).get('ok'), {k:p.get(k) for k in ('max_abs_logit_err','loss_abs_err','dense_grad_rel_err','rows_grad_rel_err','adam_rows_rel_err')})" || tail -3 $O/line_$f.err; done
