"""profiles/TAG_raycast.md from the outputs of tools/round4_raycast.sh TAG (merged into gpurun_out/): kernel time, the work
of the rays (dfa_tsdf_raycast_tally), SURVEY 8(d) bytes, HBM fetch / write counters, L2 and vector-L1 hit rates, SQ
activity.   usage: python tools/raycast_summary.py r04"""
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
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def counters(cfg):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, "%s_raycast_pmc_%s_p*" % (tag, cfg), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "raycast_points" in r["Kernel_Name"] or "raycast_depth" in r["Kernel_Name"]:
                k = ("points" if "points" in r["Kernel_Name"] else "depth", r["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
    return {k: s / n for k, (n, s) in acc.items()}


def stats(cfg):
    f = glob.glob(os.path.join(out, "%s_raycast_stats_%s" % (tag, cfg), "**", "*kernel_stats.csv"), recursive=True)
    res = {}
    if f:
        for r in csv.DictReader(open(f[0])):
            for v in ("points", "depth"):
                if "raycast_%s_kernel" % v in r["Name"]:
                    res[v] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return res


def line(cfg):
    log = os.path.join(out, "%s_raycast_stats_%s.log" % (tag, cfg))
    for ln in open(log):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


with open(os.path.join(root, "profiles", "%s_raycast.md" % tag), "w") as g:
    g.write("# Raycast (`dfa_tsdf_raycast_points` / `_depth`) at BASELINE sizes — %s, commit %s\n\n" % (tag, commit))
    g.write("`tools/round4_raycast.sh`: `rocprofv3 --kernel-trace --stats` and one `--pmc` set per pass over `tools/raycast_probe.py`\n"
            "(one fused frame, then both variants from the integration pose).  Work = the counters of `dfa_tsdf_raycast_tally` over the\n"
            "same rays; SURVEY 8(d) bytes = fetches x 4 B + image bytes out.  FETCH_SIZE / WRITE_SIZE in KiB per dispatch (gathers of 4 bytes:\n"
            "uncorrected); hit rates = TCC_HIT / (TCC_HIT + TCC_MISS) and 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES.\n\n")
    for cfg in ("C2", "C4"):
        d, c, st = line(cfg), counters(cfg), stats(cfg)
        if not d:
            continue
        w = d["work"]
        g.write("## %s\n\n" % cfg)
        g.write("rays entering the volume %d, march fetches %d (%.1f per ray), hits %d, trilinear fetches %d, distinct voxels %d "
                "(%.1f MB), distinct 64-byte lines %d (%.1f MB), distinct 128-byte lines %d (%.1f MB)\n\n"
                % (w["rays_entered"], w["march_fetches"], w["march_fetches"] / max(1, w["rays_entered"]), w["hits"], w["trilinear_fetches"],
                   w["unique_voxels"], 4e-6 * w["unique_voxels"], w.get("unique_lines_64B", 0), 64e-6 * w.get("unique_lines_64B", 0),
                   w.get("unique_lines_128B", 0), 128e-6 * w.get("unique_lines_128B", 0)))
        g.write("| variant | rocprof avg [us] (calls) | hipEvent avg [us] | SURVEY 8(d) bytes [MB] | GB/s of those | FETCH_SIZE [KiB] | WRITE_SIZE [KiB] | L2 hit rate | vector-L1 hit rate | SQ_WAIT_ANY / SQ_WAVE_CYCLES | VALU instructions |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for v in ("points", "depth"):
            e = d[v]
            hit, miss = c.get((v, "TCC_HIT_sum"), 0), c.get((v, "TCC_MISS_sum"), 0)
            tot, req = c.get((v, "TCP_TOTAL_CACHE_ACCESSES_sum"), 0), c.get((v, "TCP_TCC_READ_REQ_sum"), 0)
            g.write("| %s | %.1f (%d) | %.1f | %.1f | %.0f | %.0f | %.0f | %.2f | %.2f | %.2f | %.2e |\n"
                    % (v, st.get(v, (0, 0))[1], st.get(v, (0, 0))[0], 1e3 * e["avg_launch_ms"], 1e-6 * e["survey_bytes_per_launch"], e["achieved"],
                       c.get((v, "FETCH_SIZE"), 0), c.get((v, "WRITE_SIZE"), 0), hit / max(1, hit + miss), 1 - req / max(1, tot),
                       c.get((v, "SQ_WAIT_ANY"), 0) / max(1, c.get((v, "SQ_WAVE_CYCLES"), 0)), c.get((v, "SQ_INSTS_VALU"), 0)))
        g.write("\n")
print("wrote profiles/%s_raycast.md" % tag)
