#!/bin/bash
# Moving average on the bench shape under the A/B switches of launch_moving_average_t (dev tool, round 3).
run() { echo -n "$1: "; env $1 timeout -k 10 120 python bench.py --workload mavg --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['frac'],4), round(d['step_ms']['mean'],4), round(d['step_ms']['median'],4), round(d['step_ms']['min'],4), d['parity']['ok'] if 'parity' in d else '')"; }
for v in "$@"; do run $v; done
