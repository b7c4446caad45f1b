#!/bin/bash
# sample the GPU's clocks / power / temperature twice a second while bench.py runs (is the 1.34 vs 1.43 ms state of the
# scan kernel a clock state?): profiles/clock_watch.sh <runs>
out=gpurun_out/clock_watch.txt; : > $out
( while true; do date +%s.%N >> $out; rocm-smi -d 0 --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)" >> $out; sleep 0.5; done ) &
W=$!
for i in $(seq 1 ${1:-4}); do
  date +%s.%N > gpurun_out/cw_start_$i.txt
  python bench.py --no-extras --no-cpu-baseline > gpurun_out/cw_$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/cw_$i.json')); print('run $i', d['value'], d['stage_ms_per_launch']['scan'])"
done
kill $W
