#!/bin/bash
# Round 4 evidence for the final tree, run inside ONE gpurun call:
#   kernel trace + stats (headline, long, mixed), then the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate
#   passes, --pmc only: no tracing domain next to counters) for the same three workloads.  The bench's full result of every
#   run goes to gpurun_out/r4_*_detail.json (--detail); its last stdout line is the compact contract line.
# usage: bash profiles/collect_r4.sh     -> gpurun_out/r4_{stats,pmc}_*; `python profiles/make_traffic_json.py <wl> 4` turns them into profiles/r04_*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
keys() { if [ "$1" = mixed ]; then echo "--index-keys 950000000"; fi; }  # the mixed stream is configs[4]'s: against the 950 M-key union table
for wl in short long mixed; do
  ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --workload $wl $(keys $wl) --detail gpurun_out/r4_stats_${wl}_detail.json"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_stats_$wl -o trace -- python3 bench.py $ARGS \
    > gpurun_out/r4_stats_$wl.json 2> gpurun_out/r4_stats_$wl.err
  echo "stats $wl done"
done
for wl in short long mixed; do
  ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload $wl $(keys $wl)"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r4_pmc_${wl}_fetch -o pmc -- python3 bench.py $ARGS --detail gpurun_out/r4_pmc_${wl}_fetch_detail.json \
    > gpurun_out/r4_pmc_${wl}_fetch.json 2> gpurun_out/r4_pmc_${wl}_fetch.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d gpurun_out/r4_pmc_${wl}_write -o pmc -- python3 bench.py $ARGS --detail gpurun_out/r4_pmc_${wl}_write_detail.json \
    > gpurun_out/r4_pmc_${wl}_write.json 2> gpurun_out/r4_pmc_${wl}_write.err
  echo "pmc $wl done"
done
