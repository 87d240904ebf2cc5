#!/bin/bash
# tests/ab_run.sh "<variants>" "<configs>": phases of the bench step per variant built by tests/ab_build.sh
for c in $2; do for v in $1; do
  AEC_AMD_LIB=$PWD/build/ab/$v/libaec.so.0 timeout 120 python3 bench.py --config $c --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c', '$v', d['value'], d['phases_ms'])"
done; done
