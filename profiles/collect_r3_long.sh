#!/bin/bash
# Round 3: where the long-read step's time goes.  Timing-only builds of the scan kernel (DCN_EXP bits, results wrong) built
# on the box, then the long workload (configs[2] whole, 1.5 Gbp per batch) through each, twice, alternating.
# usage (inside gpurun): bash profiles/collect_r3_long.sh > gpurun_out/r3_long_split.txt
set -e
for v in "e2 -DDCN_EXP=2" "e1 -DDCN_EXP=1" "e256 -DDCN_EXP=256" "e512 -DDCN_EXP=512" "e1024 -DDCN_EXP=1024" "e1280 -DDCN_EXP=1280"; do
  set -- $v; name=$1; shift
  bash profiles/build_variant.sh $name "$@" > gpurun_out/build_$name.log 2>&1 &
done
wait
echo "variants built"
bash profiles/run_variants.sh "--workload long --steps 12" base e2 e1 e256 e512 e1024 e1280 base e2 e1 e256 e512 e1024 e1280
