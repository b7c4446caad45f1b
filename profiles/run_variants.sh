#!/bin/bash
# time the scan stage for each experiment variant (the full result is read from bench.py's --detail file; stdout carries the compact line): profiles/run_variants.sh "<bench args>" name1 name2 ...
ARGS=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="$GRAFT_REPO_ROOT/deacon-server_amd/lib/variants/libdeacon_hip_$v.so"; fi
  DCN_LIB_PATH=$lib timeout -k 10 300 python bench.py $ARGS --no-cpu-baseline --no-extras --detail gpurun_out/var_$v.json > gpurun_out/var_$v.line 2> gpurun_out/var_$v.err || echo "$v FAILED"
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/var_{v}.json"))
    st = d['stage_ms_per_launch']
    print(f"{v:18s} value={d['value']:.0f} ms/step={d['ms_per_step']:.3f} scan={st['scan']:.3f} distinct={st['distinct']:.3f} plan={st['plan']:.3f} kept={d['kept_fraction']:.4f} decisions_only={d['decisions_only']['value']:.0f} scan={d['decisions_only']['scan_ms_per_launch']:.3f}")
except Exception as e:
    print(v, "no result", e)
PY
done
