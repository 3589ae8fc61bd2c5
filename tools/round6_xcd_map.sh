#!/bin/bash
# do the assembly kernels (s6_assemble2 north-star, assemble_kernel reference mode) get faster when
# neighbouring nodes run on the same XCD — workgroup b -> node (b mod 8) D/8 + b/8, a contiguous range of the (spatially
# coherent) node order per XCD, so that the records a node shares with its neighbours meet in ONE L2 — and what do the
# FETCH_SIZE counters say?  DFA_XCD_MAP=0/1 in the development library; rocprofv3 kernel statistics and one --pmc pass each.
# Outputs gpurun_out/$1_xcd_*; tools/round6_xcd_summary.py writes profiles/$1_xcd_map.md from them.
tag=${1:-r06}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export DFA_LIB_PATH=$root/dynfu_amd/libdynfu_amd_dev.so
lite="--no-cpu-baseline --no-northstar --no-pipelined-probe --no-live-depth --no-end-to-end --no-other-configs --no-multi-sequence --no-raycast --no-rccl-selfcheck --repeats 1"
ns="--mode northstar --no-cpu-baseline --no-rccl-selfcheck --gn-tol 0 --repeats 1"
for m in ${XCD_MODES:-0 1 2}; do   # 2: the contiguous eighths of the MORTON order of the nodes (north-star assembly only)
  export DFA_XCD_MAP=$m
  for cfg in C3 C4; do
    steps=6; [ $cfg == C4 ] && steps=4
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_xcd_stats_ns_${cfg}_$m -o k -- python3 $root/bench.py $ns --config $cfg --steps $((steps*3)) --warmup 3 > $out/${tag}_xcd_stats_ns_${cfg}_$m.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_xcd_pmc_ns_${cfg}_$m -o k -- python3 $root/bench.py $ns --config $cfg --steps $steps --warmup 2 > $out/${tag}_xcd_pmc_ns_${cfg}_$m.log 2>&1
    [ $m == 2 ] && continue
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_xcd_stats_ref_${cfg}_$m -o k -- python3 $root/bench.py $lite --config $cfg --steps $((steps*3)) --warmup 3 > $out/${tag}_xcd_stats_ref_${cfg}_$m.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_xcd_pmc_ref_${cfg}_$m -o k -- python3 $root/bench.py $lite --config $cfg --steps $steps --warmup 2 > $out/${tag}_xcd_pmc_ref_${cfg}_$m.log 2>&1
  done
done
find $out -path "*${tag}_xcd_stats_*" -name "*kernel_trace.csv" -delete
find $out -path "*${tag}_xcd_*" -name "*agent_info.csv" -delete
du -sh $out | tail -1
