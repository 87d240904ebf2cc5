#!/bin/bash
# Collects what profiles/<round>/ holds (run on the GPU box from the repo root):
#   tests/collect_profiles.sh <outdir-under-gpurun_out>
# bench lines of every configuration, rocprofv3 kernel statistics of the same command, HBM traffic (PMC,
# separate passes) per configuration, SQ counters of the C2 kernels, the VALU issue-rate microbenchmark, index / small-call / chunked-SZIP benches, the sharded path on one GPU.
OUT=$PWD/gpurun_out/$1; R=$PWD
mkdir -p $OUT
for c in c2 c3 c5 typical; do
  python3 bench.py --config $c > $OUT/bench_${c}_4GiB.json 2> $OUT/bench_${c}.err
done
python3 bench.py --shard-path --no-extras > $OUT/bench_c2_4GiB_shard_path.json 2>> $OUT/bench_c2.err
python3 bench.py --overlap --no-extras --no-cpu-baseline > $OUT/bench_c2_4GiB_overlap.json 2>> $OUT/bench_c2.err
( cd /tmp && export TMPDIR=/tmp
  for c in c2 c3 c5 typical; do
    timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -- python3 $R/bench.py --config $c --no-cpu-baseline --no-extras > $OUT/bench_${c}_4GiB_under_rocprofv3.json 2> /dev/null
    f=$(find $OUT/stats_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_${c}_4GiB.csv
  done )
for c in c2 c3 c5 typical; do
  bash tests/prof_traffic.sh $1/traffic_$c 4096 $c > /dev/null 2>&1
  cp $OUT/traffic_$c/traffic.json $OUT/traffic_${c}_4GiB.json 2>/dev/null
done
bash tests/prof_pmc2.sh $1/sq_c2 4096 > /dev/null 2>&1
cp $OUT/sq_c2/summary.txt $OUT/pmc_sq_c2_4GiB.txt 2>/dev/null
[ -x build/valu_rate ] && timeout 120 build/valu_rate > $OUT/valu_issue_rate.txt 2>&1
[ -x build/store_pattern ] && timeout 120 build/store_pattern > $OUT/store_pattern.txt 2>&1
python3 tests/bench_index.py --config c2 --size-mib 1 64 1024 > $OUT/bench_index.txt 2>&1
python3 tests/bench_index.py --config c5 --size-mib 64 1024 >> $OUT/bench_index.txt 2>&1
python3 tests/bench_index.py --config c3 --size-mib 1 8 64 1024 >> $OUT/bench_index.txt 2>&1
python3 tests/bench_index.py --config typical --size-mib 1 8 64 1024 >> $OUT/bench_index.txt 2>&1
# index + segment starts + decode of the bare stream (the device API the ABI decode and bench decode_bare use)
for c in c2 c3 c5 typical; do
  s="64 1024 4096"; [ $c = typical ] && s="64 1024"
  python3 tests/bench_index.py --config $c --decode --size-mib $s 2>&1 | grep -v amdgpu >> $OUT/bench_index_decode.txt
done
# kernel statistics and HBM traffic of the index pass itself (tests/bench_index.py), per configuration
bash tests/prof_index.sh $1/idx 1024 > /dev/null 2>&1
for c in c2 c5 c3 typical; do cp $OUT/idx/kernel_stats_index_$c.csv $OUT/kernel_stats_index_$c.csv 2>/dev/null; done
( cd /tmp && export TMPDIR=/tmp
  for c in c2 c3; do for m in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $m --kernel-trace --output-format csv -d $OUT/idxpmc_${c}_$m -- python3 $R/tests/bench_index.py --config $c --size-mib 1024 > /dev/null 2>&1
  done; done )
python3 tests/pmc_summary.py "$OUT/idxpmc_c2_*/runc/*counter_collection.csv" > $OUT/traffic_index_c2_1GiB.txt 2>&1
python3 tests/pmc_summary.py "$OUT/idxpmc_c3_*/runc/*counter_collection.csv" > $OUT/traffic_index_c3_1GiB.txt 2>&1
# SQ counters of the index kernels (window tables: C2; trunk + coalescing walks: C3), one counter set per pass
for c in c2 c3; do
  bash tests/prof_index_pmc.sh $1/idxsq_$c $c 1024 > /dev/null 2>&1
  cp $OUT/idxsq_$c/summary_$c.txt $OUT/pmc_sq_index_${c}_1GiB.txt 2>/dev/null
done
# phase stamps of the window-table kernel (tuning build: AEC_S2_PROF)
for c in c2 c5; do
  AEC_AMD_LIB=$R/libaec_amd/lib/tuning/libaec.so.0 AEC_S2_PROF=1 python3 tests/bench_index.py --config $c --size-mib 1024 2>&1 | grep -v amdgpu > $OUT/k_spec2_phases_$c.txt
done
python3 tests/bench_index_mixed.py 2>&1 | grep -v amdgpu > $OUT/bench_index_mixed.txt
python3 tests/bench_abi_small.py > $OUT/bench_abi_small.txt 2>&1
for c in c2 c3 c5 typical; do python3 tests/bench_abi_large.py --config $c 2>&1 | grep -v amdgpu >> $OUT/bench_abi_large.txt; done
python3 tests/bench_sz_chunks.py 2>&1 | grep -v amdgpu > $OUT/bench_sz_chunks.txt
python3 tests/bench_sz_chunks.py --narrow 2>&1 | grep -v amdgpu >> $OUT/bench_sz_chunks.txt
python3 tests/bench_short_rsi.py 2>&1 | grep -v amdgpu > $OUT/bench_short_rsi.txt
python3 tests/bench_short_rsi.py --size-kib 64 --rsi 1 4 16 32 64 128 2>&1 | grep -v amdgpu >> $OUT/bench_short_rsi.txt
python3 tests/bench_short_rsi.py --size-mib 1 2>&1 | grep -v amdgpu >> $OUT/bench_short_rsi.txt
python3 tests/bench_short_rsi.py --edges 2>&1 | grep -v amdgpu > $OUT/bench_short_rsi_edges.txt
python3 tests/bench_degenerate.py --size-mib 64 2>&1 | grep -v amdgpu > $OUT/bench_degenerate.txt
# kernel trace of small one-shot decodes (which kernels a 64 KiB call is made of) and the wave-per-RSI decoder's phases
bash tests/prof_small.sh $1/small > /dev/null 2>&1
python3 tests/trace_windows.py $(find $OUT/small -name "*kernel_trace.csv" | head -1) 5 25 > $OUT/small_call_kernels_c5.txt 2>&1
AEC_AMD_LIB=$R/libaec_amd/lib/tuning/libaec.so.0 AEC_DW_PROF=1 python3 tests/bench_abi_small.py --chunk-kib 64 --reps 3 2>&1 | grep -v amdgpu > $OUT/k_decode_wave_phases_c5.txt
bash tests/prof_short_rsi.sh $1/shortrsi 16 > /dev/null 2>&1
cp $OUT/shortrsi/kernel_stats_short_rsi.csv $OUT/kernel_stats_short_rsi.csv 2>/dev/null
timeout 600 python3 tests/fuzz_index_gpu.py --cases 300 --seed 5001 --time 2>&1 | grep -A25 "slowest" > $OUT/fuzz_index_slowest_shapes.txt
python3 -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu.log
rm -f $OUT/out.txt $OUT/summary.txt
rm -rf $OUT/small $OUT/shortrsi $OUT/sq_c2 $OUT/stats_* $OUT/traffic_c2 $OUT/traffic_c3 $OUT/traffic_c5 $OUT/traffic_typical $OUT/idx $OUT/idxpmc_* $OUT/idxsq_*
