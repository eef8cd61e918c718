#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the three CPU oracles (sanitizers run on the CPU build only).
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -ffp-contract=off -fopenmp -fPIC -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o /tmp/libppo_oracle_asan.so oracle/ppo_oracle.c oracle/a2c_oracle.c oracle/dqn_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" CRL_ORACLE_SO=/tmp/libppo_oracle_asan.so \
    python -m pytest tests/test_oracle.py tests/test_a2c_oracle.py tests/test_dqn_oracle.py tests/test_golden.py tests/test_golden_widen.py -q -m "not gpu" -p no:cacheprovider
