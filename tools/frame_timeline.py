"""one frame of a rocprofv3 kernel trace as a timeline (start, end, duration, gap to the previous kernel of the same
queue): python tools/frame_timeline.py gpurun_out/prof_x/c2_kernel_trace.csv [frame-from-the-end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "compute_dists" in r["Kernel_Name"] or "grid_setup" in r["Kernel_Name"]]
# a frame begins at the grid set-up of its graph build (or the dists kernel, whichever comes first)
frames = [i for i, r in enumerate(rows) if "grid_setup" in r["Kernel_Name"] or "grid_build_one" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = frames[-back - 1], frames[-back]
t0 = int(rows[a]["Start_Timestamp"])
prev = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    gap = s - prev.get(q, s)
    prev[q] = e
    print("%8.1f %8.1f  dur %7.1f gap %6.1f q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, gap / 1e3, q, r["Kernel_Name"][:60]))
