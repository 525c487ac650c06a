#!/bin/bash
# dev tool: instruction counts of the library's overlap-save wave kernel against the ablation tool's (on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_os_cmp; rm -rf $OUT; mkdir -p $OUT
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/lib -- python bench.py --workload fir_os --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/lib.err
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/tool -- ./scripts/os_wave_ablate > /dev/null 2> $OUT/tool.err
python scripts/pmc_summary.py fir_os_wave $OUT/lib
python scripts/pmc_summary.py os_pipe $OUT/tool
