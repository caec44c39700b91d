#!/bin/bash
# round 6, call 30: the pre-split-weights variant again (swizzle by m >> 2), this time with the SQ counters next to the timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c30; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 --model AutoInt --no-cpu-baseline > $O/autoint.json 2> $O/autoint.err
python - <<'PY'
import json
j=[json.loads(l) for l in open('gpurun_out/r6c30/autoint.json') if l.startswith('{')][-1]
print('autoint', round(j['ms_per_step']*1e3,1), 'us', round(j['value']/1e6,3), j['step_us'].get('repeat_step_us'), 'parity', (j.get('parity') or {}).get('ok'))
PY
bash tools_prof.sh r6c30_autoint --steps 50 --warmup 10 --model AutoInt --no-parity | head -3 | cut -c1-110
bash tools_pmc.sh r6c30_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --model AutoInt --steps 10 --warmup 5 --no-parity > gpurun_out/r6c30_a.txt 2>&1
bash tools_pmc.sh r6c30_b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" --model AutoInt --steps 10 --warmup 5 --no-parity > gpurun_out/r6c30_b.txt 2>&1
bash tools_pmc.sh r6c30_c "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" --model AutoInt --steps 10 --warmup 5 --no-parity > gpurun_out/r6c30_c.txt 2>&1
for f in a b c; do grep -A4 "k_autoint_bwd_w\|k_autoint_fwd" gpurun_out/r6c30_$f.txt | grep -v "^--" | head -10; done
rm -rf gpurun_out/r6c30_a gpurun_out/r6c30_b gpurun_out/r6c30_c gpurun_out/r6c30_autoint/*trace*
