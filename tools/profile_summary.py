"""Turns the outputs of tools/round3_profile.sh TAG (merged into gpurun_out/) into the tracked evidence under profiles/:
  profiles/TAG_kernel_stats_<name>.csv   rocprofv3 --kernel-trace --stats per-kernel table (dfa kernels + anything above 0.5 %)
  profiles/TAG_pmc_<name>.md             FETCH_SIZE / WRITE_SIZE per dispatch per kernel (two passes)
  profiles/TAG_bench_lines.jsonl         the bench lines the profiled commands printed
usage: python tools/profile_summary.py r03a"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


lines = []
for d in sorted(glob.glob(os.path.join(out, tag + "_stats_*")) + glob.glob(os.path.join(out, tag + "_trace_hostseq_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len(tag) + 7:]
    f = find(d, "kernel_stats.csv")
    if not f:
        continue
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "dfa::" in r["Name"] or float(r["Percentage"]) >= 0.5]
    with open(os.path.join(prof, "%s_kernel_stats_%s.csv" % (tag, name)), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in keep:
            w.writerow([r["Name"].replace("(anonymous namespace)::", "")[:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    j = os.path.join(out, "%s_stats_%s.json" % (tag, name))
    if os.path.exists(j) and os.path.getsize(j) > 2:
        lines.append(json.dumps(dict(command=name, commit=commit, line=json.loads(open(j).read()))))
    else:
        log = os.path.join(out, "%s_stats_%s.log" % (tag, name))
        if os.path.exists(log):
            for ln in open(log):
                if ln.startswith("{"):
                    lines.append(json.dumps(dict(command=name, commit=commit, line=json.loads(ln))))
if lines:
    open(os.path.join(prof, "%s_bench_lines.jsonl" % tag), "w").write("\n".join(lines) + "\n")


def load(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


names = sorted({os.path.basename(d)[len(tag) + 5:].rsplit("_", 2)[0] for d in glob.glob(os.path.join(out, tag + "_pmc_*_FETCH_SIZE"))})
for name in names:
    ff = find(os.path.join(out, "%s_pmc_%s_FETCH_SIZE" % (tag, name)), "counter_collection.csv")
    fw = find(os.path.join(out, "%s_pmc_%s_WRITE_SIZE" % (tag, name)), "counter_collection.csv")
    if not (ff and fw):
        continue
    f, w = load(ff), load(fw)
    with open(os.path.join(prof, "%s_pmc_%s.md" % (tag, name)), "w") as g:
        g.write("# HBM traffic per dispatch — `%s`, %s (commit %s)\n\n" % (name, tag, commit))
        g.write("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, --kernel-trace only (tools/round3_profile.sh).  Unit: KiB per\n"
                "dispatch, mean over the dispatches.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced\n"
                "read stream — the last column applies that x2 (right for streaming reads; for 4-16 byte gathers the uncorrected column is\nthe better estimate).\n\n")
        g.write("| kernel | dispatches | FETCH_SIZE [KiB] | WRITE_SIZE [KiB] | (FETCH + WRITE) [MB] | (2 x FETCH + WRITE) [MB] |\n|---|---|---|---|---|---|\n")
        for kn in sorted(f, key=lambda n: -(f[n][1] + w.get(n, [0, 0.0])[1])):
            if "dfa::" not in kn:
                continue
            nf, sf = f[kn]
            nw, sw = w.get(kn, [0, 0.0])
            mf, mw = sf / max(nf, 1), sw / max(nw, 1)
            g.write("| `%s` | %d | %.1f | %.1f | %.3f | %.3f |\n" % (kn.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80], nf, mf, mw,
                                                                  (mf + mw) * 1024 / 1e6, (2 * mf + mw) * 1024 / 1e6))
# idle-gap tables and stage clocks of the adaptor's sequence (tools/hostseq_trace.sh)
for mode in ("ref", "northstar"):
    for src, dst in (("%s_frame_gaps_%s.md" % (tag, mode), "%s_frame_gaps_hostseq_%s.md" % (tag, mode)),):
        f = os.path.join(out, src)
        if os.path.exists(f) and os.path.getsize(f) > 10:
            open(os.path.join(prof, dst), "w").write("(rocprofv3 kernel trace of dynfu_amd/host/build/sequence_bench, 512^3, mode %s, commit %s; "
                                                     "tools/frame_gaps.py)\n\n" % (mode, commit) + open(f).read())

# ---- profiles/traffic.json: HBM bytes per launch of the kernels bench.py prices, straight from the counter passes above
# (no hand-copied numbers).  key "<config>/<kernel key>" -> bytes, formula, source file, commit.  Entries whose pass did not
# run this time are carried over from the existing file (their own source / commit stay with them).
KEYS = {  # pmc run name -> config, then kernel-name fragment -> (key, formula)
    "c2": ("C2", {"integrate_runs_kernel<true": ("fused_integrate", "2F+W"), "pcg_paired_kernel": ("pcg", "F+W")}),
    "ref_c3": ("C3", {"integrate_runs_kernel<true": ("fused_integrate", "2F+W"), "pcg_mb_step_kernel": ("pcg", "F+W"), "pcg_team_kernel": ("pcg", "F+W")}),
    "ref_c4": ("C4", {"integrate_runs_kernel<true": ("fused_integrate", "2F+W"), "pcg_mb_step_kernel": ("pcg", "F+W"), "pcg_team_kernel": ("pcg", "F+W")}),
    "raycast_C2": ("C2", {"raycast_points_kernel": ("raycast_points", "F+W"), "raycast_depth_kernel": ("raycast_depth", "F+W")}),
    "raycast_C4": ("C4", {"raycast_points_kernel": ("raycast_points", "F+W"), "raycast_depth_kernel": ("raycast_depth", "F+W")}),
    "ns_c2": ("C2", {"s6_assemble2_kernel": ("s6_assemble", "F+W"), "s6_linearise_kernel": ("s6_linearise", "2F+W"), "s6_pcg_step_kernel": ("s6_pcg_step", "F+W")}),
    "ns_c3": ("C3", {"s6_assemble2_kernel": ("s6_assemble", "F+W"), "s6_linearise_kernel": ("s6_linearise", "2F+W"), "s6_pcg_step_kernel": ("s6_pcg_step", "F+W")}),
    "ns_c4": ("C4", {"s6_assemble2_kernel": ("s6_assemble", "F+W"), "s6_linearise_kernel": ("s6_linearise", "2F+W"), "s6_pcg_step_kernel": ("s6_pcg_step", "F+W")}),
}
tj = os.path.join(prof, "traffic.json")
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
for name, (config, kmap) in KEYS.items():
    ff = find(os.path.join(out, "%s_pmc_%s_FETCH_SIZE" % (tag, name)), "counter_collection.csv")
    fw = find(os.path.join(out, "%s_pmc_%s_WRITE_SIZE" % (tag, name)), "counter_collection.csv")
    if not (ff and fw):
        continue
    f, w = load(ff), load(fw)
    for frag, (key, formula) in kmap.items():
        hits = [kn for kn in f if frag in kn]
        if not hits:
            continue
        kn = max(hits, key=lambda n: f[n][1])
        mf, mw = f[kn][1] / max(f[kn][0], 1), w.get(kn, [0, 0.0])[1] / max(w.get(kn, [1, 0.0])[0], 1)
        nbytes = ((2 * mf if formula == "2F+W" else mf) + mw) * 1024.0
        traffic["%s/%s" % (config, key)] = dict(bytes_per_launch=round(nbytes, 1), fetch_kib=round(mf, 1), write_kib=round(mw, 1),
                                                formula={"2F+W": "2 x FETCH_SIZE + WRITE_SIZE (wide read streams: the gfx950 correction of MI355X_MICROARCH.md)",
                                                         "F+W": "FETCH_SIZE + WRITE_SIZE (gathers of 4-80 bytes: uncorrected)"}[formula],
                                                source="profiles/%s_pmc_%s.md" % (tag, name), commit=commit, dispatches=f[kn][0],
                                                kernel=kn.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80])
json.dump(traffic, open(tj, "w"), indent=1, sort_keys=True)
print("wrote", sorted(os.listdir(prof))[-14:])
