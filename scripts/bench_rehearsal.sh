#!/bin/bash
# The N > 1 code paths of bench.py rehearsed on ONE GPU: torch.distributed.run with one rank and DSP_BENCH_INIT_DIST=1
# (RCCL set-up, rank check, barriers, reductions, the distributed FFT's self-verification).  Writes gpurun_out/rehearsal_*.json
set -e
cd "$GRAFT_REPO_ROOT"
export DSP_BENCH_INIT_DIST=1
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
timeout -k 10 240 $L bench.py --gpus 1 --steps 5 --warmup 2 --workload fft_dist --dist-exchange rccl --no-cpu-baseline > gpurun_out/rehearsal_fft_dist_rccl.json
timeout -k 10 240 $L bench.py --gpus 1 --steps 5 --warmup 2 --workload fft_dist --dist-exchange torch --no-cpu-baseline > gpurun_out/rehearsal_fft_dist_torch.json
timeout -k 10 240 $L bench.py --gpus 1 --steps 5 --warmup 2 --workload iir_scan --scaling strong --no-cpu-baseline > gpurun_out/rehearsal_iir_scan_strong.json
timeout -k 10 240 $L bench.py --gpus 1 --steps 5 --warmup 2 --workload fir_os --scaling strong --no-cpu-baseline > gpurun_out/rehearsal_fir_os_strong.json
# the driver's default command under the launcher: the line carries `dist` from child processes (run_dist_children) ...
timeout -k 10 300 $L bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/rehearsal_default_torchrun.json
# ... and still goes out, with dist.error, when the children crash or hang
DSP_BENCH_DIST_CHILD_TEST=crash timeout -k 10 240 $L bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rehearsal_default_child_crash.json
DSP_BENCH_DIST_CHILD_TEST=hang timeout -k 10 240 $L bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --dist-budget 20 > gpurun_out/rehearsal_default_child_hang.json
