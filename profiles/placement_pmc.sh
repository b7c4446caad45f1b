#!/bin/bash
# What differs between a fast and a slow placement of the same buffers?  profiles/placement_probe2.py (contexts, batches and the
# table created anew in turn inside ONE process; its stdout gives the scan kernel's HIP-event time per trial) under
# `rocprofv3 --pmc` (counters only, no tracing domains): per-dispatch counters of the counting kernel, to be set beside the
# trial they belong to.  Address-translation counters are taken if this rocprofv3 lists them.
# usage (GPU box): [PMC="counter ..."] profiles/placement_pmc.sh <out_prefix under gpurun_out/> [trials]
OUT=$1; T=${2:-3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail > gpurun_out/${OUT}_avail.txt 2>&1
UTCL=$(grep -o "TCP_UTCL1_[A-Z_]*MISS[A-Z_]*\|TCP_UTCL1_REQUEST" gpurun_out/${OUT}_avail.txt | sort -u | head -3 | tr '\n' ' ')
echo "translation counters: $UTCL"
timeout -k 10 400 rocprofv3 --pmc ${PMC:-TCC_MISS TCC_HIT $UTCL} --output-format csv -d gpurun_out/${OUT}_pmc -o pmc -- python3 profiles/placement_probe2.py $T > gpurun_out/${OUT}_probe.txt 2> gpurun_out/${OUT}_probe.err
tail -n 20 gpurun_out/${OUT}_probe.txt
