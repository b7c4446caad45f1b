#!/bin/bash
# Round-2 evidence run on the GPU box: the driver's bench command (every config + host path), a kernel trace of the
# headline workload, the PMC passes (separate runs, --pmc only) and the long-read kernel trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err && echo "bench ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_trace -o bench -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2_trace.json 2> gpurun_out/r2_trace.err && echo "trace ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_trace_long -o bench -- python3 bench.py --workload long --steps 12 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r2_trace_long.json 2> gpurun_out/r2_trace_long.err && echo "long trace ok"
profiles/collect_pmc.sh r2_pmc --no-extras
