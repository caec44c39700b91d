#!/bin/bash
# round 4, call 10: the split-bf16 CIN forward that never forms Z
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c10
O=gpurun_out/r4c10
timeout 600 python -m pytest tests/test_x3_gpu.py tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_reference_models_gpu.py -q -k "cin or xdeepfm or CIN or fgcnn" 2>&1 | tail -12 > $O/t_cin.txt
python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline > $O/line_x3.json 2> $O/line_x3.err
DT_CIN_FWD_Z=1 python bench.py --model xDeepFM --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $O/line_x3_z.json 2> $O/line_x3_z.err
bash tools_prof.sh r4c10_x3 --model xDeepFM --steps 20 --warmup 5 --no-parity > $O/stats_x3.txt 2>&1
tail -n 5 $O/t_cin.txt | cut -c1-300; sed -n 1,12p $O/stats_x3.txt | grep -v elementwise
for f in x3 x3_z; do python - <<PY
import json
try:
    j=json.loads([l for l in open('$O/line_$f.json') if l.startswith('{')][-1])
    p=j.get('parity') or {}
    print('$f', round(j['value']/1e6,3),'M rows/s', round(j['ms_per_step']*1e3,1),'us', round(j['step_us']['median'],1), j['roofline'].get('frac_of_f32_mfma_peak'), p.get('ok'), {k:(p.get('uniform') or {}).get(k) for k in ('max_abs_logit_err','dense_grad_rel_err','dense_grad_l2_rel_err','rows_grad_rel_err','relu_units_near_kink')})
except Exception as e:
    print('$f', 'ERR', e); print(open('$O/line_$f.err').read()[-1500:])
PY
done
timeout 900 python -X faulthandler -m pytest tests -v -m gpu -x > $O/t_all_verbose.txt 2>&1
grep -n "passed\|failed" $O/t_all_verbose.txt | tail -3; grep -n "Fatal\|Segmentation\|Abort\|Memory access fault" $O/t_all_verbose.txt | head; grep -n "PASSED\|FAILED\|ERROR" $O/t_all_verbose.txt | tail -3; grep -n -A12 "Fatal Python" $O/t_all_verbose.txt | cut -c1-200 | head -40
