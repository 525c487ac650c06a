#!/bin/bash
# usage: scripts/pmc_icache.sh <workload>...   (on the GPU box) -- instruction-cache requests / hits / misses of the bench kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for W in "$@"; do
  OUT=gpurun_out/pmc_icache/$W; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  python - $OUT $W <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if m.get("SQC_ICACHE_REQ", 0) > 1e6:
        print(sys.argv[2], k, {c: f"{x:.4g}" for c, x in m.items()}, "miss share %.4f" % (m.get("SQC_ICACHE_MISSES", 0) / m["SQC_ICACHE_REQ"]))
PY
done
