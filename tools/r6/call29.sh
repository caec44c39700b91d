#!/bin/bash
# round 6, call 29: where the attention kernels' wave cycles go (parked / issue-stalled / active, LDS conflicts, MFMA busy)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools_pmc.sh r6c29_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --model AutoInt --steps 10 --warmup 5 --no-parity > gpurun_out/r6c29_a.txt 2>&1
bash tools_pmc.sh r6c29_b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" --model AutoInt --steps 10 --warmup 5 --no-parity > gpurun_out/r6c29_b.txt 2>&1
bash tools_pmc.sh r6c29_c "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" --model AutoInt --steps 10 --warmup 5 --no-parity > gpurun_out/r6c29_c.txt 2>&1
for f in a b c; do grep -A5 "k_autoint_bwd_w\|k_autoint_fwd" gpurun_out/r6c29_$f.txt | head -14; tail -3 gpurun_out/r6c29_$f.log | cut -c1-200; done
rm -rf gpurun_out/r6c29_a gpurun_out/r6c29_b gpurun_out/r6c29_c
