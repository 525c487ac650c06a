#!/bin/bash
# On the GPU box: matrix-core vs vector-ALU direct FIR, kernel times (rocprofv3 --kernel-trace --stats) and
# utilisation counters (separate --pmc pass).  Summaries -> gpurun_out/pmc_fir/ (copy into profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_fir; rm -rf $OUT; mkdir -p $OUT
for W in fir_direct fir_direct_valu; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/stats -- python bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$W.bench.json 2> $OUT/$W.stats.err
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/$W/pmc -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/$W.pmc.err
  python scripts/pmc_summary.py fir_direct $OUT/$W/pmc | tee $OUT/$W.pmc.txt
done
