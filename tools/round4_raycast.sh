#!/bin/bash
# Round-4 evidence for the raycast (GPU box): kernel statistics and the FETCH_SIZE / WRITE_SIZE / L2 hit-miss / vector-L1
# counter passes (one --pmc set per pass, --kernel-trace only) of tools/raycast_probe.py at C2 (512^3, VGA) and C4
# (1024^3, 720p).  usage: bash tools/round4_raycast.sh TAG ; outputs under gpurun_out/TAG_raycast_*
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in C2 C4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_raycast_stats_$cfg -o k -- python3 $R/tools/raycast_probe.py $cfg 30 > $out/${tag}_raycast_stats_$cfg.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_raycast_pmc_${cfg}_p$i -o k --output-format csv -- python3 $R/tools/raycast_probe.py $cfg 6 > $out/${tag}_raycast_pmc_${cfg}_p$i.log 2>&1
  done
done
find $out -path "*${tag}_raycast_*" -name "*kernel_trace.csv" -delete
find $out -path "*${tag}_raycast_*" -name "*agent_info.csv" -delete
du -sh $out | tail -1
