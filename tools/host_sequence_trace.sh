#!/bin/bash
# dev: rocprofv3 kernel trace of the C++ adaptor's DynFusion::operator() over a synthetic PNG sequence
# usage (GPU box): bash tools/host_sequence_trace.sh C1 6 tag
cfg=${1:-C1}; n=${2:-6}; tag=${3:-hostseq}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
dir=$(mktemp -d)
python3 - "$cfg" "$n" "$dir" <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from PIL import Image
from dynfu_amd import synth, build as B
B.build_cpp_tests()
cfg = synth.CONFIGS[sys.argv[1]]; n = int(sys.argv[2]); d = sys.argv[3]
os.makedirs(d + "/depth"); os.makedirs(d + "/color")
for f in range(n):
    Image.fromarray(synth.depth_frame(cfg, f)).save("%s/depth/frame-%06d.depth.png" % (d, f))
    Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save("%s/color/frame-%06d.color.png" % (d, f))
open(d + "/dim", "w").write(str(cfg["dim"]))
PY
export DFA_SEQ_DIM=$(cat $dir/dim) DFA_SEQ_FRAME_MS=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o seq -- $GRAFT_REPO_ROOT/tests/cpp/build/test_host_io sequence $dir > $out/prof_$tag.log 2>&1
tail -3 $out/prof_$tag.log
find $out/prof_$tag -name "*.csv" | head
