#!/usr/bin/env python3
"""profiles/<tag>_xcd_map.md from the outputs of tools/round6_xcd_map.sh: per assembly kernel and configuration, the mean
launch duration (rocprofv3 --stats) and the mean FETCH_SIZE per dispatch (--pmc pass) with workgroups in launch order
(DFA_XCD_MAP=0) and with a contiguous node range per XCD (DFA_XCD_MAP=1)."""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
KERNELS = ("s6_assemble2_kernel", "s6_linearise_kernel", "assemble_kernel", "linearise_kernel")  # ("::" + name: exact kernels)


def stats(d):
    res = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if ("::" + k) in r["Name"]:
                    res[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return res


def fetch(d):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in KERNELS:
                if ("::" + k) in r["Kernel_Name"]:
                    acc[k][0] += 1
                    acc[k][1] += float(r["Counter_Value"])
    return {k: v[1] / max(1, v[0]) for k, v in acc.items()}


lines = ["# Contiguous node range per XCD in the assembly kernels (VERDICT r05 item 3)", "",
         "`tools/round6_xcd_map.sh`: development library, `DFA_XCD_MAP=1` maps workgroup b to node (b mod 8) D/8 + b/8 — the eight",
         "XCDs each take a contiguous eighth of the node order (the bench's nodes follow a spiral over the surface: spatially coherent),",
         "so that a vertex record shared by neighbouring nodes is fetched into ONE L2.  Launch time: rocprofv3 --stats mean; FETCH_SIZE:",
         "mean per dispatch of a separate --pmc pass (KiB as the counter reports it; x 2 for wide read streams per MI355X_MICROARCH.md).", "",
         "| mode | config | kernel | launch order: us | FETCH [KiB] | range per XCD: us | FETCH [KiB] | time ratio | FETCH ratio |",
         "|---|---|---|---|---|---|---|---|---|"]
for mode in ("ns", "ref"):
    for cfg in ("C3", "C4"):
        for alt in (1, 2):
            s0, s1 = (stats(os.path.join(out, "%s_xcd_stats_%s_%s_%d" % (tag, mode, cfg, m))) for m in (0, alt))
            f0, f1 = (fetch(os.path.join(out, "%s_xcd_pmc_%s_%s_%d" % (tag, mode, cfg, m))) for m in (0, alt))
            for k in KERNELS:
                if k in s0 and k in s1:
                    lines.append("| %s | %s | `%s`%s | %.1f (%d) | %.0f | %.1f (%d) | %.0f | %.3f | %.3f |" % (
                        "north-star" if mode == "ns" else "reference", cfg, k, " Morton order" if alt == 2 else "", s0[k][1], s0[k][0],
                        f0.get(k, float("nan")), s1[k][1], s1[k][0], f1.get(k, float("nan")), s1[k][1] / s0[k][1],
                        f1.get(k, float("nan")) / max(1e-9, f0.get(k, float("nan")))))
open(os.path.join(root, "profiles", "%s_xcd_map.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
