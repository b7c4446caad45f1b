#!/bin/bash
# Round-1 evidence run on the GPU box: bench (with cpu_baseline), kernel trace, PMC passes, counter calibration,
# other workloads, host-path timing.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err && echo "bench ok"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1_trace -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1_trace.json 2> gpurun_out/r1_trace.err && echo "trace ok"
profiles/collect_pmc.sh r1_pmc
[ -x profiles/microbench/probe_patterns ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o profiles/microbench/probe_patterns profiles/microbench/probe_patterns.hip
# FETCH_SIZE calibration on a known scattered pattern: 64 M random 16-byte reads of a 17 GB table = 64 M sectors
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1_calib -o pmc -- profiles/microbench/probe_patterns 31 64000000 > gpurun_out/r1_calib.txt 2>&1 && echo "calib ok"
for wl in paired long; do
  timeout -k 10 400 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1_bench_$wl.json 2> gpurun_out/r1_bench_$wl.err && echo "$wl ok"
done
timeout -k 10 300 python profiles/host_path_bench.py > gpurun_out/r1_host_path.txt 2>&1; cat gpurun_out/r1_host_path.txt
