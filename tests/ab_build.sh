#!/bin/bash
# A/B builds of libaec.so.0 with extra compiler flags (kernel variants behind macros):
#   tests/ab_build.sh <name> "<extra flags>" [file.hip ...]     -> build/ab/<name>/libaec.so.0
# only the named sources are recompiled with the flags (default aec_dec.hip); run a variant with
#   AEC_AMD_LIB=$PWD/build/ab/<name>/libaec.so.0 python3 bench.py ...
set -e
name=$1; flags=$2; shift 2; files=${@:-aec_dec.hip}
R=$(cd $(dirname $0)/.. && pwd); O=$R/build/ab/$name; mkdir -p $O
make -s -C $R/libaec_amd/csrc > /dev/null
objs=""
for o in aec_enc aec_dec aec_idx aec_region aec_shard aec_gpu aec_abi; do
  src=""; for f in $files; do [ "${f%.*}" = "$o" ] && src=$f; done
  if [ -n "$src" ]; then
    x=""; [ "${src##*.}" = "cpp" ] && x="-x hip"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function $flags $x -c $R/libaec_amd/csrc/$src -o $O/$o.o
    objs="$objs $O/$o.o"
  else
    objs="$objs $R/build/obj/$o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,libaec.so.0 -o $O/libaec.so.0 $objs
echo $O/libaec.so.0
