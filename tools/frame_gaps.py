"""Where a DynFusion::operator() frame spends the time its kernels do NOT account for: from a rocprofv3
kernel trace of dynfu_amd/host/build/sequence_bench, per frame — from one compute_dists launch to the next — the summed
kernel time, the time the device is busy (union of the kernels' intervals) and every idle gap above a threshold with the
kernels on either side of it (a gap is a host synchronisation, host work between launches, or launch latency).

usage: python tools/frame_gaps.py <kernel_trace.csv> [min_gap_us=8] [frames-to-skip=4]"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def short(n):
    n = n.replace("void ", "").replace("dfa::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:48]


starts = [i for i, r in enumerate(rows) if "compute_dists_kernel" in r["Kernel_Name"]]
frames = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][skip:]
agg = collections.defaultdict(lambda: [0, 0.0])
tot = []
for a, b in frames:
    fr = rows[a:b]
    t0, t1 = int(fr[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr) / 1e3
    busy, cur_end = 0.0, t0
    for r in fr + [rows[b]]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_end:
            gap = (s - cur_end) / 1e3
            if gap >= min_gap:
                prev = max((q for q in fr if int(q["End_Timestamp"]) <= s), key=lambda q: int(q["End_Timestamp"]), default=None)
                key = "%s -> %s" % (short(prev["Kernel_Name"]) if prev else "(frame start)", short(r["Kernel_Name"]))
                agg[key][0] += 1
                agg[key][1] += gap
        if r is not rows[b]:
            busy += max(0, e - max(s, cur_end)) / 1e3
            cur_end = max(cur_end, e)
    tot.append(((t1 - t0) / 1e3, ksum, busy, len(fr)))
n = len(frames)
print("# %d frames (the first %d skipped): per frame, microseconds" % (n, skip))
print("frame (compute_dists to compute_dists) %.0f | kernels summed %.0f | device busy %.0f | idle %.0f | launches %.0f" %
      (tuple(sum(t[i] for t in tot) / n for i in (0, 1, 2)) + (sum(t[0] - t[2] for t in tot) / n, sum(t[3] for t in tot) / n)))
print("\n| idle gap >= %.0f us between | per frame | mean us | us per frame |\n|---|---|---|---|" % min_gap)
for k, (c, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %.2f | %.1f | %.1f |" % (k, c / n, s / c, s / n))
small = sum(t[0] - t[2] for t in tot) / n - sum(s for c, s in agg.values()) / n
print("| (gaps below %.0f us: launch-to-launch latency) | | | %.1f |" % (min_gap, small))
