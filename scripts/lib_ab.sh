#!/bin/bash
# usage: scripts/lib_ab.sh <bench args...>: the same bench command on dsp_amd/lib/libdsp_gpu.so and on libdsp_gpu_prev.so (a
# build of the previous source kept beside it), alternating, on one box (dev tool)
cd "$GRAFT_REPO_ROOT/dsp_amd/lib" && cp libdsp_gpu.so new.so && cp libdsp_gpu_prev.so old.so && cd "$GRAFT_REPO_ROOT"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d.get('cold'); print('$1', 'frac', round(d['roofline']['frac'],4), 'mean', round(d['step_ms']['mean'],4), 'median', round(d['step_ms']['median'],4), 'min', round(d['step_ms']['min'],4), ('cold %.4f' % c['frac']) if c else '')"; }
for rep in 1 2 3; do
  for which in new old; do
    cp dsp_amd/lib/$which.so dsp_amd/lib/libdsp_gpu.so
    timeout -k 10 200 python bench.py "$@" 2>/dev/null | show $which || exit 1
  done
done
cp dsp_amd/lib/new.so dsp_amd/lib/libdsp_gpu.so
