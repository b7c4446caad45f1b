#!/bin/bash
# same-box A/B of library variants: profiles/run_ab.sh name1 name2 ...   (short and long workloads, twice each)
bash profiles/run_variants.sh "" "$@" "$@"
bash profiles/run_variants.sh "--workload long --steps 12" "$@"
bash profiles/run_variants.sh "--workload paired --steps 12" "$@"
