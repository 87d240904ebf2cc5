#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs on the CPU (GPU sanitizers are not available on
# the pool): the per-lane arithmetic shared with the kernels (aec_lane.h, aec_spec.h, aec_spec2.h, aec_trunk.h, aec_small.h, aec_region.h, through
# the emulators tests/emul/*.cpp) and the oracle.  Builds instrumented copies over the normal test libraries, runs the
# CPU tests that use them and puts the normal builds back.
#   bash tests/sanitize_cpu.sh            (from the repo root)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
T=$(mktemp -d); trap 'cp $T/*.so.emul tests/emul/_build/ 2>/dev/null; for f in tests/emul/_build/*.so.emul; do [ -f "$f" ] && mv $f ${f%.emul}; done; [ -f $T/oracle.so ] && cp $T/oracle.so oracle/_build/libaec_oracle.so; rm -rf $T' EXIT
python3 -m pytest tests/test_lane_emul.py tests/test_trunk_emul.py tests/test_small_emul.py tests/test_region_emul.py tests/test_oracle.py -q -x > /dev/null     # (normal builds exist)
SAN="-O1 -g -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined"
for n in libemul libtrunk_emul libsmall_emul libregion_emul; do cp tests/emul/_build/$n.so $T/$n.so.emul; done
cp oracle/_build/libaec_oracle.so $T/oracle.so
g++ $SAN -std=c++17 -Wno-unknown-pragmas -o tests/emul/_build/libemul.so tests/emul/emul.cpp
g++ $SAN -std=c++17 -Wno-unknown-pragmas -o tests/emul/_build/libtrunk_emul.so tests/emul/trunk_emul.cpp
g++ $SAN -std=c++17 -Wno-unknown-pragmas -I include -o tests/emul/_build/libsmall_emul.so tests/emul/small_emul.cpp
g++ $SAN -std=c++17 -Wno-unknown-pragmas -I include -o tests/emul/_build/libregion_emul.so tests/emul/region_emul.cpp
gcc $SAN -o oracle/_build/libaec_oracle.so oracle/aec_oracle.c
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    python3 -m pytest tests/test_lane_emul.py tests/test_trunk_emul.py tests/test_small_emul.py tests/test_region_emul.py tests/test_oracle.py -q -x
# the host threads of the batch entry points (libaec_amd/csrc/aec_pool.h) under ThreadSanitizer
g++ -O1 -g -fsanitize=thread -std=c++17 -pthread tests/c/pool_test.cpp -o $T/pool_test_tsan
TSAN_OPTIONS=die_after_fork=0 $T/pool_test_tsan
