#!/bin/bash
# Runs the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not
# available on this pool): the known-answer and second-opinion tests, then every oracle stage over the edge-case
# and seeded-sweep shapes the GPU parity tests use.  Any report fails the run.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
export STM_ORACLE_SO=$PWD/oracle/libstm_oracle_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
python -m pytest tests/test_oracle_kat.py tests/test_oracle_second_opinion.py -x -q -p no:cacheprovider
python -u tools/oracle_sweep.py
