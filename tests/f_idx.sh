cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O
for c in c2 c5; do timeout 200 python3 tests/bench_index.py --config $c --size-mib 64 1024 > $O/idx_$c.txt 2>&1; done
export AEC_AMD_LIB=$PWD/libaec_amd/lib/tuning/libaec.so.0
AEC_S2_PROF=1 timeout 200 python3 tests/bench_index.py --config c2 --size-mib 1024 > $O/prof_c2.txt 2>&1
AEC_S2_FAST=0 AEC_S2_PROF=1 timeout 200 python3 tests/bench_index.py --config c2 --size-mib 1024 > $O/prof_c2_slow.txt 2>&1
AEC_S2_VERIFY=1 timeout 200 python3 tests/bench_index.py --config c2 --size-mib 64 > $O/verify_c2.txt 2>&1
AEC_S2_VERIFY=1 timeout 200 python3 tests/bench_index.py --config c5 --size-mib 64 > $O/verify_c5.txt 2>&1
tail -n 3 $O/*.txt
