#!/bin/bash
# Runs on the GPU box (through gpurun): for every bench workload named on the command line
#   rocprofv3 --kernel-trace --stats            (kernel durations of the same bench command)
#   rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE     (HBM traffic, separate passes)
#   rocprofv3 --pmc <SQ counters>               (two passes: where the waves' cycles go, LDS conflicts)
# Results land in gpurun_out/prof_$ROUND/<workload>/ ; scripts/profile_collect.py turns them into profiles/.
set -e
ROUND=${ROUND:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROOT=gpurun_out/prof_$ROUND
B="--no-cpu-baseline --no-secondary"
for W in "$@"; do
  OUT=$ROOT/$W; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --workload $W --steps 20 --warmup 3 $B > $OUT/bench.json 2> $OUT/stats.err
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --workload $W --steps 3 --warmup 1 $B > /dev/null 2> $OUT/fetch.err
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --workload $W --steps 3 --warmup 1 $B > /dev/null 2> $OUT/write.err
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sqa -- python bench.py --workload $W --steps 3 --warmup 1 $B > /dev/null 2> $OUT/sqa.err
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/sqb -- python bench.py --workload $W --steps 3 --warmup 1 $B > /dev/null 2> $OUT/sqb.err
  echo "profiled $W"
done
