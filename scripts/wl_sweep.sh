#!/bin/bash
# usage: scripts/wl_sweep.sh <workload> [VAR=value ...]: the bench workload under each environment setting (dev tool)
W=$1; shift
run() { echo -n "$W $1: "; env $1 timeout -k 10 120 python bench.py --workload $W --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['frac'],4), round(d['step_ms']['mean'],4), round(d['step_ms']['median'],4), round(d['step_ms']['min'],4))"; }
for v in "$@"; do run $v; done
