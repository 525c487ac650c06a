#!/bin/bash
# usage: scripts/pmc_tool.sh <binary> <out-name>   (on the GPU box) -- SQ counters of every kernel a dev tool launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$1; OUT=gpurun_out/pmc_tool_$2
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a -- $B > $OUT/a.out 2> $OUT/a.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -- $B > $OUT/b.out 2> $OUT/b.err || exit 1
python - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"{out}/summary.txt", "w") as fo:
    for k in acc:
        v = {c: sum(x) / len(x) for c, x in acc[k].items()}
        d = sum(dur[k]) / len(dur[k])
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        line = f"{k}\n   dur {d:.0f} us waves {v.get('SQ_WAVES',0):.0f} clock~{v.get('SQ_BUSY_CYCLES',0)/d/32:.0f} MHz"
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU"):
            if c in v: line += f" {c[3:]}/wc={v[c]/wc:.3f}"
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES"):
            if c in v: line += f" {c[3:]}={v[c]:.4g}"
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/a $OUT/b
