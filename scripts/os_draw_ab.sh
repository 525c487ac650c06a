#!/bin/bash
# Overlap-save FIR on C3 with its block pairs DRAWN from counters (k pairs per draw) or dealt out in advance (0): dev tool
bash scripts/wl_sweep.sh fir_os DSP_GPU_OS_WAVE_DRAW=0 DSP_GPU_OS_WAVE_DRAW=1 DSP_GPU_OS_WAVE_DRAW=2 DSP_GPU_OS_WAVE_DRAW=4 DSP_GPU_OS_WAVE_DRAW=0 DSP_GPU_OS_WAVE_DRAW=1 DSP_GPU_OS_WAVE_DRAW=2 DSP_GPU_OS_WAVE_DRAW=3
