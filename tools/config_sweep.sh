#!/bin/bash
# The BASELINE configurations other than the headline on the current build (parity-test cases, not bench lines): frames/s only.
OUT=${1:-gpurun_out/configs}
mkdir -p $OUT
for cfg in "--stages 1" "--stages 2" "--stages 259" "--stages 3" "--disp 128" "--height 2160 --width 3840 --disp 256"; do
  name=$(echo $cfg | tr -d ' -')
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 $cfg > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $cfg"
  python3 -c "import json,sys; d=json.load(open('$OUT/$name.json')); print('%-45s %8.1f frames/s %8.3f ms' % ('$cfg', d['value'], d['ms_per_step']))"
done
