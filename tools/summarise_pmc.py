"""gpurun_out/pmc_<round>_<workload>/ (tools/pmc_workload.sh) -> profiles/<round>_<workload>_pmc.json: per kernel, the mean
counter values per launch, with the derived figures the roofline discussion uses."""
import csv, glob, json, os, re, sys
R, WL = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (R, WL))
sha = open(os.path.join(OUT, "source_sha.txt")).read().strip()
per = {}
for f in glob.glob(os.path.join(OUT, "g*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if not re.search(r"k_sweep|k_wsweep|k_wfinalize|k_wreduce|k_hub_frag|k_fold_finalize|k_fe_|k_em_|k_nonedge|k_row_sums|k_moments", name):
            continue
        k = re.sub(r"\(.*", "", name).replace("void sbmbp::", "")
        per.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        per[k].setdefault("_vgpr", set()).add(row.get("VGPR_Count", row.get("Arch_VGPR_Count", "")))
        per[k].setdefault("_lds", set()).add(row.get("LDS_Block_Size", ""))
out = {"workload": WL, "source_sha": sha,
       "command": "rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 bench.py --workload %s --steps 5 --warmup 2 --no-cpu-baseline --no-converge (one pass per counter group; tools/pmc_workload.sh)" % WL,
       "notes": "SQ_* cycle counters are in quad-cycles summed over waves / SEs as rocprofv3 reports them; FETCH_SIZE and WRITE_SIZE in KiB (on gfx950 wide coalesced reads are tallied at half: MI355X_MICROARCH.md HBM)",
       "kernels": {}}
for k, c in sorted(per.items()):
    e = {"vgpr": sorted(x for x in c.pop("_vgpr") if x), "lds_bytes": sorted(x for x in c.pop("_lds") if x),
         "counters": {n: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for n, v in sorted(c.items())}}
    m = {n: d["per_launch_mean"] for n, d in e["counters"].items()}
    d = {}
    if "SQ_WAVE_CYCLES" in m and m.get("SQ_WAVES"):
        d["wave_cycles_per_wave_x4"] = 4 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"]
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if n in m:
                d[n.lower() + "_share_of_wave_cycles"] = m[n] / m["SQ_WAVE_CYCLES"]
        if "SQ_BUSY_CYCLES" in m:
            d["mean_waves_in_flight_per_simd_estimate"] = m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CYCLES"] if m["SQ_BUSY_CYCLES"] else None
    if m.get("SQ_INSTS_VALU") and m.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) > 0:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if m.get("SQ_INSTS_MFMA") is not None and m.get("SQ_WAVES"):
        d["mfma_insts_per_wave"] = m["SQ_INSTS_MFMA"] / m["SQ_WAVES"]
    if "FETCH_SIZE" in m:
        d["fetch_bytes_as_counted"] = m["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in m:
        d["write_bytes"] = m["WRITE_SIZE"] * 1024.0
    e["derived"] = d
    out["kernels"][k] = e
dst = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (R, WL.lower()))
json.dump(out, open(dst, "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, e["vgpr"], json.dumps(e["derived"]))
print("wrote", dst)
