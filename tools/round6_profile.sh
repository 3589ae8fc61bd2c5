#!/bin/bash
# Round-6 evidence on the GPU box (successor of tools/round5_profile.sh: + the raycast's FETCH_SIZE / WRITE_SIZE passes, so that
# profiles/traffic.json holds no figure carried over from a hand-kept table): rocprofv3 kernel statistics and the FETCH_SIZE / WRITE_SIZE counter passes (one counter
# per pass, --kernel-trace only; the program directly behind `--`) of the default bench (C2), of the reference-mode bench at
# C3 / C4, of the north-star bench at C2 / C3 / C4, and the kernel trace + idle-gap table of the C++ adaptor's sequence.
# Outputs under gpurun_out/$1_*; `python tools/profile_summary.py $1` turns them into the tracked files under profiles/
# (kernel tables, per-dispatch traffic, bench lines, and profiles/traffic.json — the table bench.py's `traffic` fields read).
tag=${1:-r06}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
lite="--repeats 1 --no-cpu-baseline --no-northstar --no-pipelined-probe --no-live-depth --no-end-to-end --no-other-configs --no-multi-sequence --no-raycast --no-rccl-selfcheck"
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
ns="--mode northstar --repeats 1 --no-cpu-baseline --no-rccl-selfcheck"
# north-star: the default (stopping rule on: launches behind a stop return at entry and dilute every per-launch mean) for the
# bench lines, and --gn-tol 0 (every launch does its work) for per-launch durations and traffic
if [ "$2" != "ns-fixed-only" ]; then
prof c2 $lite
prof ref_c3 $lite --config C3 --steps 12
prof ns_c2 $ns --config C2 --steps 40
prof ns_c3 $ns --config C3 --steps 30
prof ns_c4 $ns --config C4 --steps 10 --warmup 4
pmc c2 $lite --steps 10 --warmup 2
pmc ref_c3 $lite --config C3 --steps 6 --warmup 2
pmc ref_c4 $lite --config C4 --steps 4 --warmup 2
fi
prof ns_c2_fixed $ns --config C2 --steps 40 --gn-tol 0
prof ns_c3_fixed $ns --config C3 --steps 30 --gn-tol 0
pmc ns_c2 $ns --config C2 --steps 6 --warmup 2 --gn-tol 0
pmc ns_c3 $ns --config C3 --steps 4 --warmup 2 --gn-tol 0
pmc ns_c4 $ns --config C4 --steps 3 --warmup 2 --gn-tol 0
if [ "$2" == "ns-fixed-only" ]; then exit 0; fi
# the raycast (reported separately): kernel statistics + the two traffic passes at C2 and C4
for cfg in C2 C4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_raycast_$cfg -o k -- python3 $root/tools/raycast_probe.py $cfg 30 > $out/${tag}_stats_raycast_$cfg.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_raycast_${cfg}_$c -o k -- python3 $root/tools/raycast_probe.py $cfg 6 > $out/${tag}_pmc_raycast_${cfg}_$c.log 2>&1
  done
done
# the adaptor's sequence (DynFusion::operator(), 512^3, ~1.08 M vertices, ~8.5 k nodes): kernel statistics + idle gaps
bash $root/tools/hostseq_trace.sh $tag 14
# keep what the summary needs, drop the bulky traces of the stats runs
find $out -path "*${tag}_stats_*" -name "*kernel_trace.csv" -delete
find $out -path "*${tag}_*" -name "*agent_info.csv" -delete
du -sh $out | tail -1
