"""Condense the rocprofv3 output of tools/profile_bench.sh (gpurun_out/<tag>/) into a small, committed summary:
  profiles/<name>_kernel_stats.csv   (the --stats per-kernel table, our kernels only)
  profiles/<name>_summary.json       (avg duration, PMC counters, corrected HBM traffic per launch)
HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB and, on
gfx950, FETCH_SIZE reports half the bytes of a 16-byte-per-lane streaming read (MI355X_MICROARCH.md §HBM);
WRITE_SIZE is exact for 16-byte stores.  Usage: python tools/summarize_prof.py <tag> <name>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(path)
    if not files:
        return d
    # (gpurun merges every call's output into the same tree: the newest file is this run's)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def short(name):
    return name.replace("void ", "").split("(")[0]


def main(tag, name):
    src = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(stats)) if "mfa::" in r["Name"]]
    with open(os.path.join(ROOT, "profiles", f"{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    summary = {"source": f"tools/profile_bench.sh {tag}: calls / avg_us / min_us / max_us from `rocprofv3 --kernel-trace --stats -- "
                         "python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-sweep` (every launch, warm-up included); "
                         "the counters from separate `rocprofv3 --pmc ... -- python3 bench.py --steps 20 --warmup 5 "
                         "--no-cpu-baseline --no-sweep` passes (FETCH_SIZE | WRITE_SIZE | SQ_*), averaged over their launches",
               "kernels": {}}
    pm = {k: counters(os.path.join(src, k, "*", "*_counter_collection.csv")) for k in ("pmc_fetch", "pmc_write", "pmc_sq")}
    for r in rows:
        k = r["Name"]
        e = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
             "max_us": float(r["MaxNs"]) / 1e3}
        for grp in pm.values():
            for c, vals in grp.get(k, {}).items():
                e[c] = sum(vals) / len(vals)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes"] = 2 * e["FETCH_SIZE"] * 1024
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
            e["hbm_traffic_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            e["shader_cycles"] = cyc
            e["mfma_busy_frac_of_1024_simds"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
        summary["kernels"][short(k)] = e
    with open(os.path.join(ROOT, "profiles", f"{name}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # per-launch HBM traffic of the dominant kernels, read back by bench.py for roofline.traffic
    traffic = {"source": f"profiles/{name}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                         "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 correction)"}
    for k, e in summary["kernels"].items():
        if "hbm_traffic_bytes" in e:
            key = "prefill" if "prefill64_kernel" in k else ("decode" if "decode_split" in k else ("kvcache_packed" if "prefill_fwd_kernel" in k else None))
            if key:
                traffic[key] = e["hbm_traffic_bytes"]
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
