"""per-dispatch durations of the rulebook kernels of ONE step from a rocprofv3 --kernel-trace CSV (gpurun_out/<tag>/..._kernel_trace.csv)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = ("rb_", "order_local", "fillBufferAligned", "scan_")
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: take the last N dispatches where N = per-step count
n_fill = sum("rb_fill" in r["Kernel_Name"] for r in sel)
n_steps = max(1, sum("rb_hash_insert" in r["Kernel_Name"] for r in sel) // 1)
per = len(sel) // max(1, int(sys.argv[2]) if len(sys.argv) > 2 else 36)
for r in sel[-per:]:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
    print("%-28s %8.1f us  grid %8s wg %4s  stream %s" % (name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Stream_Id", r.get("Queue_Id", "?"))))
