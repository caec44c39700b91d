#!/bin/bash
# round 3, GPU call 27: MFMA-busy counters of the two MFMA paths (CIN, AutoInt) — separate PMC passes, --kernel-trace --pmc only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools_pmc.sh r03_pmc_mfma_xdeepfm "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --model xDeepFM --steps 5 --warmup 2 --no-parity > gpurun_out/r03_pmc_mfma_xdeepfm.txt 2>&1
bash tools_pmc.sh r03_pmc_mfma_autoint "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --model AutoInt --steps 10 --warmup 2 --no-parity > gpurun_out/r03_pmc_mfma_autoint.txt 2>&1
bash tools_pmc.sh r03_pmc_mfma_deepfm "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --steps 20 --warmup 5 --no-parity > gpurun_out/r03_pmc_mfma_deepfm.txt 2>&1
head -40 gpurun_out/r03_pmc_mfma_xdeepfm.txt | cut -c1-120
grep -A3 "autoint_bwd_w\|autoint_fwd" gpurun_out/r03_pmc_mfma_autoint.txt | cut -c1-120 | head -12
