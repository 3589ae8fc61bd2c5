#!/bin/bash
# Round-3 evidence on the GPU box: rocprofv3 kernel statistics and the FETCH_SIZE / WRITE_SIZE counter passes (one counter
# per pass, --kernel-trace only) of the default bench (C2), of the north-star bench at C2 and C3, and a kernel trace of the
# C++ adaptor's north-star sequence at 512^3.  Outputs under gpurun_out/$1_*; tools/profile_summary.py $1 turns them into
# the tracked files under profiles/.
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
lite="--no-cpu-baseline --no-northstar --no-pipelined-probe --no-live-depth --no-end-to-end"
prof() {  # name, then the bench arguments
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$name -o k -- python3 $root/bench.py "$@" > $out/${tag}_stats_$name.log 2>&1
  grep '^{' $out/${tag}_stats_$name.log | tail -1 > $out/${tag}_stats_$name.json
}
pmc() {
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${name}_$c -o k -- python3 $root/bench.py "$@" > $out/${tag}_pmc_${name}_$c.log 2>&1
  done
}
prof c2 $lite
prof ns_c2 --mode northstar --config C2 --no-cpu-baseline --steps 40
prof ns_c3 --mode northstar --config C3 --no-cpu-baseline --steps 30
pmc c2 $lite --steps 10 --warmup 2
pmc ns_c2 --mode northstar --config C2 --no-cpu-baseline --steps 6 --warmup 2
pmc ns_c3 --mode northstar --config C3 --no-cpu-baseline --steps 4 --warmup 2
# the adaptor's north-star sequence (DynFusion::operator(), 512^3, ~1.08 M vertices, ~8.5 k nodes)
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import numpy as np
from dynfu_amd import synth
cfg = synth.CONFIGS["C2"]
np.stack([synth.depth_frame(cfg, f) for f in range(12)]).astype("<u2").tofile("/tmp/frames.u16")
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_hostseq_ns -o k -- $root/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 12 512 northstar > $out/${tag}_stats_hostseq_ns.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_hostseq_ref -o k -- $root/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 12 512 ref > $out/${tag}_stats_hostseq_ref.log 2>&1
# keep what the summary needs, drop the bulky traces of the stats runs
find $out -path "*${tag}_stats_*" -name "*kernel_trace.csv" -delete
find $out -path "*${tag}_*" -name "*agent_info.csv" -delete
du -sh $out | tail -1
