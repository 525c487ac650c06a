#!/bin/bash
# usage: scripts/pmc_run.sh <workload> <kernel-substring>   (on the GPU box; writes gpurun_out/pmc_<workload>/)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=$1; K=$2; OUT=gpurun_out/pmc_$W
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/a.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/b.err
python scripts/pmc_summary.py "$K" $OUT/a $OUT/b | tee $OUT/summary.txt
