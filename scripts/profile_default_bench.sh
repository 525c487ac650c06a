#!/bin/bash
# rocprofv3 --kernel-trace --stats of the round-end bench command itself (all workloads in one process, the headline
# FFT timed last): the kernel averages there are the ones the default bench line's roofline entries must agree with.
set -e
ROUND=${ROUND:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$ROUND/bench_default; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.err
echo "profiled the default bench command"
