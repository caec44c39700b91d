#!/bin/bash
# round 6, call 31: SQ counters of the DeepFM step's four launches (where their waves' cycles go)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools_pmc.sh r6c31_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --steps 20 --warmup 5 --no-parity > gpurun_out/r6c31_a.txt 2>&1
bash tools_pmc.sh r6c31_b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" --steps 20 --warmup 5 --no-parity > gpurun_out/r6c31_b.txt 2>&1
bash tools_pmc.sh r6c31_c "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" --steps 20 --warmup 5 --no-parity > gpurun_out/r6c31_c.txt 2>&1
bash tools_pmc.sh r6c31_d "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES" --steps 20 --warmup 5 --no-parity > gpurun_out/r6c31_d.txt 2>&1
for f in a b c d; do grep -A4 "k_tower_x3\|k_wgrad_rows\|k_sparse_fwd\|k_finish_step" gpurun_out/r6c31_$f.txt | grep -v "^--"; tail -2 gpurun_out/r6c31_$f.log | cut -c1-160; done
rm -rf gpurun_out/r6c31_a gpurun_out/r6c31_b gpurun_out/r6c31_c gpurun_out/r6c31_d
