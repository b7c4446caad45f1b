#!/bin/bash
# long-read workload A/B on one box
bash profiles/run_variants.sh "--workload long --steps 12" base prev base prev
bash profiles/run_variants.sh "" base prev
