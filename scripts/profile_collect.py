#!/usr/bin/env python3
"""Turn gpurun_out/prof_<round>/<workload>/ (scripts/profile_all.sh) into the committed summaries:
profiles/<round>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
profiles/<round>_<workload>_pmc.txt            SQ counter summary of the dominant kernel (scripts/pmc_summary.py)
profiles/traffic.json                          HBM bytes per launch of the dominant kernel (FETCH_SIZE doubled per the
                                               gfx950 correction of MI355X_MICROARCH.md section HBM, WRITE_SIZE exact)
usage: profile_collect.py [round]   (default r02)"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{ROUND}")
KERNEL = {"fft": "fft_lds_kernel", "fir_direct": "fir_direct_kernel", "fir_os": "fir_os_wave2_kernel",
          "mavg": "moving_average_kernel", "iir_batch": "iir_batch_kernel", "iir_scan": "iir_scan_kernel",
          "fft_dist": "fft_column_kernel", "fft_32k": "fft_plane_kernel", "fft_real_64k": "fft_plane_kernel",
          "fft_f64": "fft64_tile_kernel", "biquad_swept": "biquad_swept_kernel",
          "biquad_swept_shared": "biquad_swept_shared_kernel"}
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime, reverse=True)
for w, key in KERNEL.items():
    d = os.path.join(SRC, w)
    if not os.path.isdir(d):
        continue
    stats = newest(os.path.join(d, "stats", "**", "*kernel_stats.csv"))
    avg_ns = calls = None
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{ROUND}_{w}_kernel_stats.csv"))
        for r in csv.DictReader(open(stats[0])):
            if key in r["Name"] and avg_ns is None:
                avg_ns = float(r["AverageNs"]); calls = int(r["Calls"])
    if os.path.isdir(os.path.join(d, "sqa")) and os.path.isdir(os.path.join(d, "sqb")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), key, os.path.join(d, "sqa"), os.path.join(d, "sqb")],
                             capture_output=True, text=True)
        if out.returncode == 0:
            with open(os.path.join(ROOT, "profiles", f"{ROUND}_{w}_pmc.txt"), "w") as f:
                f.write(f"# rocprofv3 --pmc (two passes, scripts/profile_all.sh) -- python bench.py --workload {w} --steps 3 --warmup 1; averages per dispatch\n")
                f.write("# SQ_*_CYCLES / ACTIVE / WAIT counters are in quad-cycles summed over waves; LDS bank conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n")
                f.write("# (calibration of that counter on known patterns: profiles/r02_lds_bank_conflict_calibration.txt)\n")
                f.write(out.stdout)
    vals = {}
    for kind, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fs = newest(os.path.join(d, kind, "**", "*counter_collection.csv"))[:1]
        v = [float(r["Counter_Value"]) for f in fs for r in csv.DictReader(open(f)) if key in r["Kernel_Name"] and r["Counter_Name"] == cn]
        if v:
            vals[cn] = sum(v) / len(v)
    bj = os.path.join(d, "bench.json")
    bench = json.load(open(bj)) if os.path.exists(bj) and os.path.getsize(bj) else {}
    if len(vals) == 2:
        traffic[w] = {"hbm_bytes_per_launch": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
                      "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                      "correction": "gfx950: FETCH_SIZE counts 128-B read requests at 64 B -> doubled; WRITE_SIZE exact; KiB units",
                      "kernel": key, "kernel_avg_ms_rocprof": avg_ns / 1e6 if avg_ns else None, "kernel_calls": calls,
                      "bench_launch_ms_hip_events": bench.get("roofline", {}).get("launch_ms"),
                      "round": int(ROUND[1:]), "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload {w} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"}
        print(w, round(traffic[w]["hbm_bytes_per_launch"] / 2**30, 3), "GiB", "rocprof avg ms", traffic[w]["kernel_avg_ms_rocprof"], "bench ms", traffic[w]["bench_launch_ms_hip_events"])
json.dump(traffic, open(tpath, "w"), indent=1)

# the round-end bench command itself (scripts/profile_default_bench.sh): its line, its kernel stats, every launch in order
d = os.path.join(SRC, "bench_default")
if os.path.isdir(d):
    bj = os.path.join(d, "bench.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        shutil.copy(bj, os.path.join(ROOT, "profiles", f"{ROUND}_bench_default.json"))
    stats = newest(os.path.join(d, "stats", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{ROUND}_bench_default_kernel_stats.csv"))
    trace = newest(os.path.join(d, "stats", "**", "*kernel_trace.csv"))
    if trace:
        order = ["fft_lds_kernel", "moving_average_kernel", "fir_os_wave2_kernel", "iir_batch_kernel", "iir_scan_kernel"]
        runs = {k: [] for k in order}
        for r in csv.DictReader(open(trace[0])):
            for k in order:
                if k in r["Kernel_Name"]:
                    runs[k].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        with open(os.path.join(ROOT, "profiles", f"{ROUND}_bench_default_per_launch_us.txt"), "w") as f:
            f.write("# rocprofv3 --kernel-trace of the default bench command (python bench.py --gpus 1 --steps 20 --warmup 5): duration of every\n"
                    "# launch of each memory-bound kernel in microseconds, in launch order (5 warm-up + 20 timed + 20 per-step launches per\n"
                    "# workload; the FFT twice: cold at the start of the process, sustained at its end).  Every workload shows the same shape:\n"
                    "# its first 3-4 launches are the fastest, launches 5-15 the slowest (the chip's power management answers a new load by\n"
                    "# dropping the shader clock, profiles/r03_fft_clock_ramp_after_idle.txt), then a slow recovery -- and the driver's 5 warm-up\n"
                    "# launches are exactly the fast ones.  (scripts/profile_collect.py)\n")
            for k in order:
                v = [u for _, u in sorted(runs[k])]
                med = sorted(v)[len(v) // 2] if v else 0.0
                v = [u for u in v if u > 0.25 * med]  # (small launches of the same kernel by the parity checks)
                if v:
                    f.write(f"{k}   min {min(v):.0f}  launches 6-25 (the timed ones) mean {sum(v[5:25]) / max(len(v[5:25]), 1):.0f}  last 10 mean {sum(v[-10:]) / len(v[-10:]):.0f}\n  "
                            + " ".join(f"{u:.0f}" for u in v) + "\n")
