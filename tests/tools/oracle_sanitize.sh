#!/bin/bash
# The CPU oracle suites against an ASan + UBSan build of oracle/icp_oracle.c (make -C oracle sanitize): out-of-bounds reads in the
# restatement would otherwise pass silently as "parity".  CPU only.  usage: tests/tools/oracle_sanitize.sh [pytest args]
set -e
cd "$(dirname "$0")/../.."
make -s -C oracle sanitize
export ICP_ORACLE_LIB=$PWD/oracle/_san/liboracle_san.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
python -m pytest tests/test_oracle_golden.py tests/test_oracle_ext.py -q -x -p no:cacheprovider "$@"
