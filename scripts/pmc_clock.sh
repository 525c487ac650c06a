#!/bin/bash
# usage: scripts/pmc_clock.sh <workload> <kernel-substring> ...   (on the GPU box; pairs of arguments)
# Core clock under each kernel: GRBM_GUI_ACTIVE (cycles the graphics pipe is busy) over the dispatch's duration.
# Writes gpurun_out/pmc_clock/<workload>.txt (dev tool, round 3).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_clock
while [ $# -ge 2 ]; do
  W=$1; K=$2; shift 2
  OUT=gpurun_out/pmc_clock/$W
  rm -rf $OUT
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT.err || exit 1
  python scripts/pmc_summary.py "$K" $OUT > gpurun_out/pmc_clock/$W.txt || exit 1
  python - "$W" gpurun_out/pmc_clock/$W.txt <<'PY'
import re, sys
w, f = sys.argv[1:]
t = open(f).read()
dur = float(re.search(r"dur_us\(profiled\)=([0-9.]+)", t).group(1))
g = float(re.search(r"GRBM_GUI_ACTIVE\s+([0-9.e+]+)", t).group(1))
b = float(re.search(r"SQ_BUSY_CYCLES\s+([0-9.e+]+)", t).group(1))
line = f"{w}: dur {dur:.1f} us, GRBM_GUI_ACTIVE/dur = {g/dur:.0f} cycles/us (/8 XCDs = {g/dur/8:.0f} MHz), SQ_BUSY_CYCLES/dur/32 = {b/dur/32:.0f} MHz"
print(line)
open(f, "a").write("  " + line + "\n")
PY
  rm -rf $OUT
done
