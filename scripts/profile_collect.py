#!/usr/bin/env python3
"""Turn gpurun_out/prof_r01/<workload>/ (scripts/profile_all.sh) into the committed summaries:
profiles/r01_<workload>_kernel_stats.csv and profiles/traffic.json (HBM bytes per launch of the dominant
kernel, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md section HBM)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r01")
KERNEL = {"fft": "fft_lds_kernel", "fir_direct": "fir_direct_kernel", "fir_os": "fir_overlap_save_kernel",
          "mavg": "moving_average_kernel", "iir_batch": "iir_batch_kernel", "iir_scan": "iir_scan_kernel"}
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for w, key in KERNEL.items():
    d = os.path.join(SRC, w)
    if not os.path.isdir(d):
        continue
    # gpurun merges new files into gpurun_out/ without deleting older runs: take the newest
    stats = sorted(glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"r01_{w}_kernel_stats.csv"))
        for r in csv.DictReader(open(stats[0])):
            if key in r["Name"]:
                avg_ns = float(r["AverageNs"]); calls = int(r["Calls"])
    vals = {}
    for kind, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fs = sorted(glob.glob(os.path.join(d, kind, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)[:1]
        v = [float(r["Counter_Value"]) for f in fs for r in csv.DictReader(open(f)) if key in r["Kernel_Name"] and r["Counter_Name"] == cn]
        if v:
            vals[cn] = sum(v) / len(v)
    bench = json.load(open(os.path.join(d, "bench.json"))) if os.path.getsize(os.path.join(d, "bench.json")) else {}
    if len(vals) == 2:
        traffic[w] = {"hbm_bytes_per_launch": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
                      "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                      "correction": "gfx950: FETCH_SIZE counts 128-B read requests at 64 B -> doubled; WRITE_SIZE exact; KiB units",
                      "kernel": key, "kernel_avg_ms_rocprof": avg_ns / 1e6 if stats else None, "kernel_calls": calls if stats else None,
                      "bench_launch_ms_hip_events": bench.get("roofline", {}).get("launch_ms"),
                      "round": 1, "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload {w} --steps 3 --warmup 1 --no-cpu-baseline"}
        print(w, traffic[w]["hbm_bytes_per_launch"] / 2**30, "GiB", "rocprof avg ms", traffic[w]["kernel_avg_ms_rocprof"], "bench ms", traffic[w]["bench_launch_ms_hip_events"])
json.dump(traffic, open(tpath, "w"), indent=1)
