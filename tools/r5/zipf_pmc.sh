#!/bin/bash
# Hot-row evidence: the gather kernel (k_sparse_fwd) under PMC with uniform and Zipf(1.05) ids — L2 hit / miss,
# fabric read requests, wave wait cycles.  One counter group per pass (separate --pmc runs, kernel-trace only).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/r05_zipf_pmc.txt
: > $OUT
for dist in uniform zipf; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    tag=zp_${dist}_$(echo $grp | cut -d' ' -f1)
    bash tools_pmc.sh $tag "$grp" --steps 20 --warmup 5 --no-parity --dist $dist > gpurun_out/$tag.txt 2>&1
    echo "== ids $dist | $grp" >> $OUT
    grep -A8 "k_sparse_fwd" gpurun_out/$tag.txt | head -8 >> $OUT
    grep -A6 "k_wgrad_rows" gpurun_out/$tag.txt | head -6 >> $OUT
    grep -A6 "k_finish_step" gpurun_out/$tag.txt | head -6 >> $OUT
  done
done
for dist in uniform zipf; do
  bash tools_prof.sh zp_stats_$dist --steps 100 --warmup 10 --no-parity --dist $dist | head -6 >> $OUT
done
cat $OUT | head -120
