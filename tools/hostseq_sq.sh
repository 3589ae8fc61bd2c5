#!/bin/bash
# SQ / cache counters of the kernels of the C++ adaptor's sequence (DynFusion::operator(), 512^3): rocprofv3 --pmc in passes of
# four counters, --kernel-trace only, the program directly behind `--`.  usage (GPU box): bash tools/hostseq_sq.sh TAG [mode=ref]
tag=${1:-r05}; mode=${2:-ref}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/${tag}_sq_hostseq_$mode; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$root")
import numpy as np
from dynfu_amd import synth
cfg = synth.CONFIGS["C2"]
np.stack([synth.depth_frame(cfg, f) for f in range(8)]).astype("<u2").tofile("/tmp/frames.u16")
PY
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_WR" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o k --output-format csv -- $root/dynfu_amd/host/build/sequence_bench /tmp/frames.u16 640 480 8 512 $mode > $out/p$i.log 2>&1
done
find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
python3 - <<PY
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].replace("void ", "").replace("dfa::", "").replace("(anonymous namespace)::", "").split("(")[0][:28]
        acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
ks = sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0])))[:14]
with open("$out/summary.md", "w") as g:
    g.write("| counter | " + " | ".join("%s" % k for k in ks) + " |\n|---|" + "---|" * len(ks) + "\n")
    for c in names:
        g.write("| %s | " % c + " | ".join("%.3g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else "-" for k in ks) + " |\n")
print(open("$out/summary.md").read())
PY
