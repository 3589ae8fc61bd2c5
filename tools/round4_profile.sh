#!/bin/bash
# Round-4 evidence on the GPU box (everything under gpurun_out/$1_*; tools/profile_summary.py $1 and
# tools/raycast_summary.py $1 turn it into the tracked files under profiles/):
#   kernel statistics + FETCH_SIZE / WRITE_SIZE passes of the default bench (C2), the reference-mode bench at C3 / C4, the
#   north-star bench at C2 / C3, the adaptor's sequence (DynFusion::operator(), 512^3) in both modes, and the raycast probe
#   at C2 / C4 (with L2 hit / miss and vector-L1 counters).  One --pmc set per pass, --kernel-trace only.
tag=${1:-r04}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
lite="--no-cpu-baseline --no-northstar --no-pipelined-probe --no-live-depth --no-end-to-end --no-other-configs --no-raycast"
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
prof ref_c3 --config C3 $lite --steps 40
prof ref_c4 --config C4 $lite --steps 30
prof ns_c2 --mode northstar --config C2 --no-cpu-baseline --steps 40
prof ns_c3 --mode northstar --config C3 --no-cpu-baseline --steps 30
pmc c2 $lite --steps 10 --warmup 2
pmc ref_c3 --config C3 $lite --steps 6 --warmup 2
pmc ns_c2 --mode northstar --config C2 --no-cpu-baseline --steps 6 --warmup 2
pmc ns_c3 --mode northstar --config C3 --no-cpu-baseline --steps 4 --warmup 2
find $out -path "*${tag}_stats_*" -name "*kernel_trace.csv" -delete
find $out -path "*${tag}_*" -name "*agent_info.csv" -delete
# the adaptor's sequence: names hostseq_ref / hostseq_northstar (same summary tool)
bash $root/tools/round4_hostseq.sh $tag > /dev/null 2>&1
bash $root/tools/round4_raycast.sh $tag > /dev/null 2>&1
du -sh $out | tail -1
