#!/bin/bash
# Collects the PMC passes for bench.py on the GPU box (separate passes; --pmc only, no tracing domains).
# usage: profiles/collect_pmc.sh <out_prefix under gpurun_out/> [bench args...]
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${OUT}_$name -o pmc -- python3 bench.py $ARGS > gpurun_out/${OUT}_$name.json 2> gpurun_out/${OUT}_$name.err
  echo "$name done"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT TCC_MISS
