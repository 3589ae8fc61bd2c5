#!/bin/bash
# kernel trace of a bench command reduced to a per-frame table: kernels summed by name, idle gaps (all queues together) by
# the kernels either side.  A frame = from one launch of MARKER to the next.
# usage (GPU box): bash tools/frame_table.sh OUT.md MARKER_KERNEL -- <bench.py arguments>
outf=$1; marker=$2; shift 3
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o k -- python3 $root/bench.py "$@" > $d/log 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$marker" > $outf <<'PY'
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("void ", "").replace("dfa::", "").replace("(anonymous namespace)::", "").split("(")[0][:38]
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
frames = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-40:]
gaps = collections.defaultdict(lambda: [0, 0.0]); ks = collections.defaultdict(lambda: [0, 0.0]); tot = []
for a, b in frames:
    fr = rows[a:b]; t0 = int(fr[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
    cur = t0; busy = 0
    for r in fr + [rows[b]]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur:
            prev = max((q for q in fr if int(q["End_Timestamp"]) <= s), key=lambda q: int(q["End_Timestamp"]), default=None)
            key = "%s -> %s" % (short(prev["Kernel_Name"]) if prev else "-", short(r["Kernel_Name"]))
            gaps[key][0] += 1; gaps[key][1] += (s - cur) / 1e3
        if r is not rows[b]:
            busy += max(0, e - max(s, cur)); cur = max(cur, e)
            k = short(r["Kernel_Name"]); ks[k][0] += 1; ks[k][1] += (e - s) / 1e3
    tot.append(((t1 - t0) / 1e3, busy / 1e3, len(fr)))
n = len(frames)
print("frames %d: frame %.1f us, device busy (union) %.1f, idle %.1f, launches %.1f" % (n, sum(t[0] for t in tot) / n, sum(t[1] for t in tot) / n, sum(t[0] - t[1] for t in tot) / n, sum(t[2] for t in tot) / n))
print("\n| kernel | launches / frame | us / frame | us / launch |\n|---|---|---|---|")
for k, (c, s) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:30]:
    print("| %s | %.1f | %.1f | %.1f |" % (k, c / n, s / n, s / c))
print("\n| gap between | per frame | us / frame |\n|---|---|---|")
for k, (c, s) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:20]:
    print("| %s | %.2f | %.1f |" % (k, c / n, s / n))
PY
rm -rf $d
