#!/bin/bash
# long-read workload: tile sizes on one box (run_variants.sh prints one line per run)
for tw in 512 384 256 192 128 96; do
  echo "DCN_TILE_WINDOWS=$tw"; DCN_TILE_WINDOWS=$tw bash profiles/run_variants.sh "--workload long --steps 12" base
done
