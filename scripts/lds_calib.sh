#!/bin/bash
# dev tool (GPU box): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE on patterns of known conflict degree -> profiles/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lds_calib; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- ./scripts/lds_calib > $OUT/names.txt 2> $OUT/a.err
python - <<'PY'
import csv, glob
rows = {}
for f in glob.glob("gpurun_out/lds_calib/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = [l.strip() for l in open("gpurun_out/lds_calib/names.txt") if l.strip()]
print(f"{'pattern':78s} {'conflict/active':>15s} {'active/inst':>12s}")
for (d, v), name in zip(sorted(rows.items()), names):
    print(f"{name:78s} {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):15.3f} {v['SQ_LDS_IDX_ACTIVE'] / max(v['SQ_INSTS_LDS'], 1):12.2f}")
PY
