"""Per-dispatch SQ counters of the conv kernels from a rocprofv3 --pmc pass over tools/conv_bench.py.

usage: python tools/sq_counters.py <csv dir>      (after: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python tools/conv_bench.py 0)
Prints, per distinct (kernel, grid), the mean of every counter and the ratios that say where a wave's cycles go
(MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, in quad-cycles; MFMA_BUSY in cycles)."""
import csv, glob, os, sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "conv_apply" not in name and "conv_wgrad" not in name:
            continue
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = (short, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    n = len(next(iter(cs.values())))
    line = "%-46s grid %8s lds %6s x%-3d" % (key[0][:46], key[1], key[2], n)
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_MISC"):
        if c in m:
            line += " %s %.2f" % (c[3:].replace("_INST", "").lower(), m[c] / wc)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        line += " mfma_busy/busy %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(m["SQ_BUSY_CYCLES"], 1))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        line += " bank_conf %.2f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1))
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES"):
        if c in m:
            line += " %s %.3g" % (c.replace("SQ_", "").lower(), m[c])
    print(line)
