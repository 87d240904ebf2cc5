#!/bin/bash
# The fuzzers on the build at hand (GPU box, from the repo root): tests/fuzz_all.sh <outdir-under-gpurun_out>
# index 300 + 300 (two seeds), the streaming calls 200 + 40 big, damaged streams 300, batches 150; a log per run, the last
# line of each says "mismatches: N".
O=$PWD/gpurun_out/$1; mkdir -p $O
timeout 1500 python3 tests/fuzz_index_gpu.py --cases 300 --seed 5001 2>&1 | grep -v amdgpu > $O/index_5001.log
timeout 1500 python3 tests/fuzz_index_gpu.py --cases 300 --seed 777 2>&1 | grep -v amdgpu > $O/index_777.log
timeout 1500 python3 tests/fuzz_stream_gpu.py --cases 200 --seed 7101 2>&1 | grep -v amdgpu > $O/stream_7101.log
timeout 1500 python3 tests/fuzz_stream_gpu.py --cases 40 --seed 504 --big 2>&1 | grep -v amdgpu > $O/stream_big_504.log
timeout 1500 python3 tests/fuzz_corrupt_gpu.py --cases 300 --seed 5002 2>&1 | grep -v amdgpu > $O/corrupt_5002.log
timeout 1500 python3 tests/fuzz_batch_gpu.py --cases 150 --seed 5003 2>&1 | grep -v amdgpu > $O/batch_5003.log
for f in $O/*.log; do echo "$(basename $f): $(tail -1 $f)"; done
