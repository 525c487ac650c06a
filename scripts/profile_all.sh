#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel stats + HBM traffic counters for every bench workload.
# Results land in gpurun_out/prof_r01/<workload>/ ; scripts/profile_collect.py turns them into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROOT=gpurun_out/prof_r01
for W in "$@"; do
  OUT=$ROOT/$W; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.err
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/write.err
  echo "profiled $W"
done
