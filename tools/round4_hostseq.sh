#!/bin/bash
# Round-4 evidence for the adaptor's frame (DynFusion::operator(), 512^3, ~1.08 M vertices, ~8.5 k nodes): kernel statistics
# and FETCH_SIZE / WRITE_SIZE passes of dynfu_amd/host/build/sequence_bench in both modes (the program directly behind
# `rocprofv3 ... --`).   usage: bash tools/round4_hostseq.sh TAG
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np
from dynfu_amd import synth
cfg = synth.CONFIGS["C2"]
np.stack([synth.depth_frame(cfg, f) for f in range(12)]).astype("<u2").tofile("/tmp/frames.u16")
PY
for mode in ref northstar; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_hostseq_$mode -o k -- $R/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 12 512 $mode > $out/${tag}_stats_hostseq_$mode.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_hostseq_${mode}_$c -o k -- $R/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 12 512 $mode > $out/${tag}_pmc_hostseq_${mode}_$c.log 2>&1
  done
done
find $out -path "*${tag}_*hostseq*" -name "*kernel_trace.csv" -delete
find $out -path "*${tag}_*hostseq*" -name "*agent_info.csv" -delete
du -sh $out | tail -1
