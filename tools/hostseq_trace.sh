#!/bin/bash
# kernel trace of the C++ adaptor's sequence (DynFusion::operator(), 512^3) for tools/frame_gaps.py and the kernel statistics
# usage (GPU box): bash tools/hostseq_trace.sh TAG [frames=14]
tag=${1:-r05}; n=${2:-14}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import numpy as np
from dynfu_amd import synth
cfg = synth.CONFIGS["C2"]
np.stack([synth.depth_frame(cfg, f) for f in range($n)]).astype("<u2").tofile("/tmp/frames.u16")
PY
for mode in ref northstar; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_hostseq_$mode -o k -- $root/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 $n 512 $mode > $out/${tag}_trace_hostseq_$mode.log 2>&1
  f=$(find $out/${tag}_trace_hostseq_$mode -name "*kernel_trace.csv" | head -1)
  python3 $root/tools/frame_gaps.py $f > $out/${tag}_frame_gaps_$mode.md
  find $out/${tag}_trace_hostseq_$mode -name "*kernel_trace.csv" -delete
  find $out/${tag}_trace_hostseq_$mode -name "*agent_info.csv" -delete
  DFA_HOST_PROFILE=1 $root/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 $n 512 $mode 2> $out/${tag}_host_profile_$mode.txt > /dev/null
done
