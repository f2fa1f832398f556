#!/bin/bash
# The round's profile set on the GPU box: rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only),
# SQ counter passes of the aggregation kernels.   usage: bash tools/profile_round.sh <outdir> <commit>
set -e
OUT=${1:-gpurun_out/profile_round}
COMMIT=${2:-unknown}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/ks -o f --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/ks.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o f --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/w.log 2>&1
python3 tools/pmc_traffic.py $(find $OUT/f -name '*counter_collection.csv') $(find $OUT/w -name '*counter_collection.csv') $OUT/pmc_traffic.json $COMMIT > $OUT/pmc_traffic.txt
bash tools/pmc_sq_agg.sh $OUT/sq 0
echo profile_round done
