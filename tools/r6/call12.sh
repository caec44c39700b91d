#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c12; mkdir -p $O
sed -i 's/const int wm = (!x3 || stamps) ? 0 : bf16_flag ? 1 : 2;/const int wm = 0;/' deeptables_amd/csrc/deepfm.hip
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1; tail -1 $O/build.txt
for sd in 4 5 6 7; do SEED=$sd REPS=16 timeout 900 python tools/r6/dbg_lockstep2.py > $O/lock2_wm0_$sd.txt 2>&1; echo wm0 seed $sd: $(grep -c "no table divergence" $O/lock2_wm0_$sd.txt) clean, $(grep -c "table diff" $O/lock2_wm0_$sd.txt) forks; done
