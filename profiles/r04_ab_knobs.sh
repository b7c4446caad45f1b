#!/bin/bash
# round 4, last session: the phase-B compile-time switches that were last timed BEFORE the wave scans moved to DPP (which freed
# 9 VGPRs in the counting kernel and took every spill out): software-pipelined phase B, two items per lane in the final flush,
# three waves per SIMD for the decisions-only kernel.  Variants are built on the box; runs interleaved, twice.
set -e
bash profiles/build_variant.sh pipeb -DDCN_PIPE_B=1 > gpurun_out/build_pipeb.log 2>&1
bash profiles/build_variant.sh u2 -DDCN_U_FINAL=2 > gpurun_out/build_u2.log 2>&1
bash profiles/build_variant.sh fw3 -DDCN_MIN_WAVES_FAST=3 > gpurun_out/build_fw3.log 2>&1
echo "== short (configs[1], 10 M reads per step)"
bash profiles/run_variants.sh "--steps 20 --warmup 5" base pipeb u2 fw3 base pipeb u2 fw3
echo "== long (configs[2])"
bash profiles/run_variants.sh "--workload long --steps 12 --warmup 3" base pipeb u2 base pipeb u2
