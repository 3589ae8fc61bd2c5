#!/bin/bash
# SQ instruction / activity counters of the north-star kernels at one configuration (rocprofv3 --pmc, --kernel-trace only, four
# passes of four counters), the program being tools/ns_assemble_time.py.  usage (GPU box): bash tools/sq_counters.sh TAG C3
tag=${1:-r03}; cfg=${2:-C3}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_sq_$cfg
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o k --output-format csv -- python3 $R/tools/ns_assemble_time.py $cfg > $out/p$i.log 2>&1
done
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
