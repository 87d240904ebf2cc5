#!/bin/bash
# A/B builds of libaec.so.0 with extra compiler flags (kernel variants behind macros):
#   tests/ab_build.sh <name> "<extra flags>" [file.hip ...]     -> build/ab/<name>/libaec.so.0
# only the named sources are recompiled with the flags (default aec_dec.hip: its object and its five per-block-size
# parts, -DAEC_DEC_PART=<bs>; aec_enc.hip alike); run a variant with
#   AEC_AMD_LIB=$PWD/build/ab/<name>/libaec.so.0 python3 bench.py ...
set -e
name=$1; flags=$2; shift 2; files=${@:-aec_dec.hip}
R=$(cd $(dirname $0)/.. && pwd); O=$R/build/ab/$name; mkdir -p $O
make -s -C $R/libaec_amd/csrc > /dev/null
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function"
objs=""; pids=""
for o in aec_enc aec_dec aec_idx aec_region aec_shard aec_gpu aec_abi; do
  src=""; for f in $files; do [ "${f%.*}" = "$o" ] && src=$f; done
  parts=""; [ $o = aec_dec ] && parts="DEC"; [ $o = aec_enc ] && parts="ENC"
  if [ -n "$src" ]; then
    x=""; [ "${src##*.}" = "cpp" ] && x="-x hip"
    $CC $flags $x -c $R/libaec_amd/csrc/$src -o $O/$o.o & pids="$pids $!"
    objs="$objs $O/$o.o"
    if [ -n "$parts" ]; then for b in 0 8 16 32 64; do
      $CC $flags -DAEC_${parts}_PART=$b -c $R/libaec_amd/csrc/$src -o $O/${o}_bs$b.o & pids="$pids $!"
      objs="$objs $O/${o}_bs$b.o"
    done; fi
  else
    objs="$objs $R/build/obj/$o.o"
    if [ -n "$parts" ]; then for b in 0 8 16 32 64; do objs="$objs $R/build/obj/${o}_bs$b.o"; done; fi
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-soname,libaec.so.0 -o $O/libaec.so.0 $objs
echo $O/libaec.so.0
