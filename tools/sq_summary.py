"""gpurun_out/<TAG>_sq_<CFG>/p*/…counter_collection.csv (tools/sq_counters.sh) -> profiles/<TAG>_sq_ns_<cfg>.md: mean counter
values per dispatch of the north-star kernels.  usage: python tools/sq_summary.py r03h C3"""
import collections
import csv
import glob
import os
import subprocess
import sys

tag, cfg = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "%s_sq_%s" % (tag, cfg), "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        for key in ("s6_assemble2", "s6_linearise", "s6_pcg_step", "s6_pattern"):
            if key in kn:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
names = sorted({c for v in acc.values() for c in v})
kernels = [k for k in ("s6_assemble2", "s6_linearise", "s6_pcg_step", "s6_pattern") if k in acc]
lines = ["# SQ counters per dispatch — north-star kernels at %s, %s (commit %s)" % (cfg, tag, commit), "",
         "rocprofv3 --pmc (four passes of four counters, --kernel-trace only; tools/sq_counters.sh), program tools/ns_assemble_time.py:",
         "mean over the dispatches of each kernel.  SQ_BUSY_CYCLES is summed over the 32 shader engines (/ 32 = the kernel's",
         "duration in clocks); SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count in units of 4 clocks, summed over the waves.", "",
         "| counter | " + " | ".join("`%s`" % k for k in kernels) + " |", "|---|" + "---|" * len(kernels)]
for c in names:
    lines.append("| %s | " % c + " | ".join("%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else "–" for k in kernels) + " |")
lines += ["", "Dispatches: " + ", ".join("%s %d" % (k, len(next(iter(acc[k].values())))) for k in kernels) + "."]
if cfg == "C3":
    lines += ["", "The same counters of the assembly's THIRD form (a quad of lanes per work unit, commit 8e2dbf5, 0.315 ms per launch at C3):",
              "SQ_INSTS_VALU 9.155e7, SQ_INSTS_SALU 2.91e7, SQ_INSTS_LDS 9.40e6, SQ_INSTS_VMEM_RD 2.89e6, SQ_BUSY_CYCLES 2.322e7,",
              "SQ_WAVE_CYCLES 6.795e8, SQ_ACTIVE_INST_ANY 1.533e8, SQ_ACTIVE_INST_VALU 9.17e7, SQ_WAIT_INST_ANY 5.29e7, SQ_LDS_BANK_CONFLICT 1.311e7;",
              "of its vector instructions ~9.4 M (10 %) were the packed FMAs of the products."]
open(os.path.join(root, "profiles", "%s_sq_ns_%s.md" % (tag, cfg.lower())), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
