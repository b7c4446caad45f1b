#!/bin/bash
# does a physically contiguous table (hipExtMallocWithFlags(hipDeviceMallocContiguous), DCN_TABLE_CONTIGUOUS=1) take the run-to-run
# levels out of the scan kernel's time?  Fresh processes, alternating, same box; the full result is read from --detail.
for rep in 1 2 3 4; do
  for v in plain contiguous; do
    if [ $v = contiguous ]; then export DCN_TABLE_CONTIGUOUS=1; else unset DCN_TABLE_CONTIGUOUS; fi
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --detail gpurun_out/ct_$v.json > gpurun_out/ct_$v.line 2> gpurun_out/ct_$v.err || echo "$v FAILED"
    grep -h "no contiguous allocation" gpurun_out/ct_$v.err
    python - $v <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/ct_{sys.argv[1]}.json")); r=d["roofline"]
print(f"{sys.argv[1]:11s} value={d['value']:.0f} scan={d['stage_ms_per_launch']['scan']:.3f} avg_launch={r['avg_launch_ms']:.3f} random={r.get('probe_ceiling_random_per_s',0)/1e9:.1f} replay={r.get('probe_ceiling_replay_per_s',0)/1e9:.1f} G/s index_build_s={d.get('index_build_s')}")
PY
  done
done
