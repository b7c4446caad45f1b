#!/bin/bash
# Round 3 evidence for the final tree, run inside ONE gpurun call:
#   kernel trace + stats (headline, long, mixed), then the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate
#   passes, --pmc only: no tracing domain next to counters) for the same three workloads.
# usage: bash profiles/collect_r3.sh     -> gpurun_out/r3_{stats,pmc}_*; profiles/make_traffic_json.py turns them into profiles/r03_*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
keys() { if [ "$1" = mixed ]; then echo "--index-keys 950000000"; fi; }  # the mixed stream is configs[4]'s: against the 950 M-key union table
for wl in short long mixed; do
  ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-extras --workload $wl $(keys $wl)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats_$wl -o trace -- python3 bench.py $ARGS \
    > gpurun_out/r3_stats_$wl.json 2> gpurun_out/r3_stats_$wl.err
  echo "stats $wl done"
done
for wl in short long mixed; do
  ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload $wl $(keys $wl)"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r3_pmc_${wl}_fetch -o pmc -- python3 bench.py $ARGS \
    > gpurun_out/r3_pmc_${wl}_fetch.json 2> gpurun_out/r3_pmc_${wl}_fetch.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d gpurun_out/r3_pmc_${wl}_write -o pmc -- python3 bench.py $ARGS \
    > gpurun_out/r3_pmc_${wl}_write.json 2> gpurun_out/r3_pmc_${wl}_write.err
  echo "pmc $wl done"
done
