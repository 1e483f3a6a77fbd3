"""gpurun_out/pmc/p*/ (rocprofv3 --pmc passes of tools/pmc_kernels.py, tools/gpu_pmc.sh) -> profiles/<name>.json.
Per kernel: counter averages over the last 3 launches, HBM bytes per launch with the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE is reported in KiB and counts 32-B requests of 64-B lines: x2 for the 16-B-per-lane
streaming kernels), MFMA busy share, LDS bank conflicts, L2 hit rate."""
import csv, glob, json, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
vals = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.match(r"(?:void )?(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
        if not m:
            continue
        vals[m.group(1).replace(" ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc <one set per pass> -- python tools/pmc_kernels.py (tools/gpu_pmc.sh), MI355X, averages over the "
                 "last 3 launches", "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes applies the gfx950 x2 "
                 "correction to FETCH_SIZE (MI355X_MICROARCH.md, HBM section)", "kernels": {}}
for k, cs in sorted(vals.items()):
    c = {n: sum(v[-3:]) / len(v[-3:]) for n, v in cs.items()}
    e = {"counters": {n: round(v, 1) for n, v in c.items()}}
    if "GRBM_GUI_ACTIVE" in c:
        e["kernel_cycles"] = round(c["GRBM_GUI_ACTIVE"] / 8)           # summed over the 8 XCDs
        e["kernel_us_at_2.4GHz"] = round(e["kernel_cycles"] / 2400.0, 2)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "kernel_cycles" in e:
        e["mfma_busy_frac_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["kernel_cycles"] * 1024.0), 4)   # 256 CUs x 4 SIMDs
    if "SQ_LDS_BANK_CONFLICT" in c:
        e["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
    out["kernels"][k] = e
json.dump(out, open(dst, "w"), indent=1)
print({k: {n: v for n, v in e.items() if n != "counters"} for k, e in out["kernels"].items()})
