"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family -> profiles/rNN_pmc_conv.json.

usage: python tools/pmc_summary.py FETCH_csv_dir WRITE_csv_dir out.json ["extra bench.py arguments of the passes"]
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> x1024); FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section)
prescribes for gfx950 wide coalesced reads; WRITE_SIZE is uncalibrated there."""
import csv, glob, json, os, sys

FAMILIES = {"conv_apply": ("conv_apply",), "conv_wgrad": ("conv_wgrad", "wgrad_reduce"), "rulebook": ("rb_",)}


STEPS = {}   # counter -> optimizer steps seen in that pass (27 forward BatchNorm applications per step)


def collect(d, counter):
    tot = {k: [0.0, 0] for k in FAMILIES}
    n_bn = 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            n_bn += "bn_apply<" in name
            for fam, keys in FAMILIES.items():
                if any(k in name for k in keys):
                    tot[fam][0] += float(r["Counter_Value"])
                    tot[fam][1] += 1 if "wgrad_reduce" not in name else 0
    STEPS[counter] = max(n_bn / 27.0, 1.0)
    return tot


def main():
    fd, wd, out = sys.argv[1:4]
    extra = (" " + sys.argv[4].strip()) if len(sys.argv) > 4 and sys.argv[4].strip() else ""
    f, w = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    res = {"commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline" + extra,
                        "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline" + extra],
           "units": "FETCH_SIZE / WRITE_SIZE in KB (x1024 bytes); FETCH_SIZE doubled for gfx950 wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated"}
    for fam in FAMILIES:
        n = max(f[fam][1], 1)
        fk, wk = f[fam][0] / n, w[fam][0] / max(w[fam][1], 1)
        res[fam] = {"launches": f[fam][1], "fetch_size_kb_per_launch": round(fk, 2), "write_size_kb_per_launch": round(wk, 2),
                    "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024, -3)), "note": "(2*%.2f + %.2f) * 1024" % (fk, wk),
                    "launches_per_step": round(f[fam][1] / STEPS["FETCH_SIZE"], 2),
                    "hbm_bytes_per_step": int(round((2 * f[fam][0] / STEPS["FETCH_SIZE"] + w[fam][0] / STEPS["WRITE_SIZE"]) * 1024, -3))}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
