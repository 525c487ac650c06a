#!/usr/bin/env python3
"""Average per-dispatch PMC values of one kernel from rocprofv3 counter_collection CSVs (dev tool).
usage: pmc_summary.py <kernel-substring> <dir> [<dir> ...]"""
import csv, glob, sys
from collections import defaultdict
key = sys.argv[1]
acc = defaultdict(list)
dur = []
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == list(acc)[0]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
if not acc:
    sys.exit(f"no dispatch matches {key}")
print(f"kernel~{key} vgpr={meta[0]} agpr={meta[1]} sgpr={meta[2]} lds={meta[3]} wg={meta[4]} grid={meta[5]} dur_us(profiled)={sum(dur)/max(1,len(dur)):.1f}")
v = {k: sum(x) / len(x) for k, x in acc.items()}
for k in sorted(v):
    print(f"  {k:28s} {v[k]:.4g}")
wc = v.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in v:
            print(f"  {k}/WAVE_CYCLES = {v[k]/wc:.3f}")
if "SQ_LDS_IDX_ACTIVE" in v and "SQ_LDS_BANK_CONFLICT" in v and v["SQ_LDS_IDX_ACTIVE"]:
    print(f"  LDS bank conflict share = {v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE']:.3f}")
if "SQ_BUSY_CYCLES" in v and "SQ_LDS_IDX_ACTIVE" in v:
    print(f"  LDS_IDX_ACTIVE / BUSY_CYCLES = {v['SQ_LDS_IDX_ACTIVE']/v['SQ_BUSY_CYCLES']:.3f}")
