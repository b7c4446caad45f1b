#!/bin/bash
# Round-1 evidence run on the GPU box: bench (with cpu_baseline), kernel trace, PMC passes, host-path timing.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err && echo "bench ok"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1_trace -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1_trace.json 2> gpurun_out/r1_trace.err && echo "trace ok"
profiles/collect_pmc.sh r1_pmc
timeout -k 10 300 python profiles/host_path_bench.py > gpurun_out/r1_host_path.txt 2>&1; cat gpurun_out/r1_host_path.txt
