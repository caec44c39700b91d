#!/bin/bash
# round 3, GPU call 1: the round's new parity tests + the two prepared kernel edits (kernel A's two-round-trip gather, DCN's
# unguarded coeff.Wc), measured
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_fused_gpu.py tests/test_reference_models_gpu.py tests/test_bf16_gpu.py tests/test_weights_gpu.py tests/test_checkpoint.py tests/test_autoint_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3c1_tests.txt
tail -15 gpurun_out/r3c1_tests.txt
bash tools_prof.sh r3c1_deepfm --steps 100 --warmup 10 --no-parity > gpurun_out/r3c1_stats_deepfm.txt 2>&1
bash tools_prof.sh r3c1_dcn --model DCN --steps 50 --warmup 8 --no-parity > gpurun_out/r3c1_stats_dcn.txt 2>&1
python bench.py --no-cpu-baseline --no-parity --steps 200 > gpurun_out/r3c1_line_deepfm.json 2> gpurun_out/r3c1_line_deepfm.err
python bench.py --model xDeepFM --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r3c1_line_xdeepfm.json 2> gpurun_out/r3c1_line_xdeepfm.err
python bench.py --model AutoInt --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r3c1_line_autoint.json 2> gpurun_out/r3c1_line_autoint.err
cat gpurun_out/r3c1_stats_deepfm.txt | head -12; cat gpurun_out/r3c1_stats_dcn.txt | head -10
cut -c1-400 gpurun_out/r3c1_line_deepfm.json
python - <<'PY'
import json
for m in ('xdeepfm','autoint'):
    try:
        j=json.loads(open(f'gpurun_out/r3c1_line_{m}.json').read().strip().splitlines()[-1])
        print(m, j['value'], j['roofline']['frac'], json.dumps(j.get('parity'))[:1500])
    except Exception as e: print(m,'ERR',e)
PY
