"""dev tool: mean FETCH_SIZE / WRITE_SIZE (KiB per dispatch) per kernel from two rocprofv3 --pmc passes.
    python tools/pmc_table.py gpurun_out/pmc_TAG_FETCH_SIZE/solve_counter_collection.csv gpurun_out/pmc_TAG_WRITE_SIZE/solve_counter_collection.csv"""
import csv, sys, collections


def load(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


f, w = load(sys.argv[1]), load(sys.argv[2])
print("| kernel | dispatches | FETCH_SIZE [KiB] | WRITE_SIZE [KiB] | (FETCH + WRITE) [MB] | (2 x FETCH + WRITE) [MB] |")
print("|---|---|---|---|---|---|")
for name in sorted(f, key=lambda n: -(f[n][1] + w.get(n, [0, 0.0])[1])):
    if "dfa::" not in name:
        continue
    nf, sf = f[name]
    nw, sw = w.get(name, [0, 0.0])
    mf, mw = sf / max(nf, 1), sw / max(nw, 1)
    print("| `%s` | %d | %.1f | %.1f | %.3f | %.3f |" % (name.split("(")[0].replace("void ", "")[:70], nf, mf, mw, (mf + mw) * 1024 / 1e6,
                                                        (2 * mf + mw) * 1024 / 1e6))
