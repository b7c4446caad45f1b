#!/bin/bash
# A/B of decisions-only kernel variants: profiles/run_fast_variants.sh name1 name2 ... ("base" = the in-tree library)
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="$GRAFT_REPO_ROOT/deacon-server_amd/lib/variants/libdeacon_hip_$v.so"; fi
  DCN_LIB_PATH=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || echo "$v FAILED"
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.load(open(f"gpurun_out/var_{v}.json"))
print(f"{v:8s} counting={d['value']:.0f} scan={d['stage_ms_per_launch']['scan']:.3f}  decisions_only={d['decisions_only']['value']:.0f} scan={d['decisions_only']['scan_ms_per_launch']:.3f} same={d['decisions_only']['decisions_identical_to_counting_mode']}")
PY
done
