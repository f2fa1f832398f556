#!/bin/bash
# rocprofv3 kernel stats of the other BASELINE configurations (C3 + HSLO, C4: D = 128, C5: 3840 x 2160, D = 256) and the frames/s sweep.
# usage (on the GPU box): bash tools/profile_configs.sh <outdir>
set -e
OUT=${1:-gpurun_out/prof_configs}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/c3 -o f --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --stages 259 > $OUT/c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/c4 -o f --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --disp 128 > $OUT/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/c5 -o f --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --height 2160 --width 3840 --disp 256 > $OUT/c5.log 2>&1
bash tools/config_sweep.sh $OUT/sweep > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
