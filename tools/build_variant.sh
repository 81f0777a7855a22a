#!/bin/bash
# build one library variant: tools/build_variant.sh <name> [extra hipcc flags...]  -> htscodecs_amd/variants/lib<name>.so
# (objects are rebuilt from scratch, and a failed compile fails the script: a stale object must never be linked)
set -e
cd "$(dirname "$0")/../htscodecs_amd/csrc"
name=$1; shift
mkdir -p ../variants ../../build/var_$name
rm -f ../../build/var_$name/*.o
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
objs=""; pids=""
for f in r4x16_api r4x16_host r4x16_multi r4x16_stripe r4x16_sched r4x16_decode r4x16_encode r4x16_enc_chain r4x16_enc_chain_rec; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o ../../build/var_$name/$f.o & pids="$pids $!"; objs="$objs ../../build/var_$name/$f.o"
done
/opt/rocm/bin/hipcc $FLAGS -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c r4x16_enc_chain_pk.hip -o ../../build/var_$name/r4x16_enc_chain_pk.o & pids="$pids $!"
for p in $pids; do wait $p || { echo "build_variant: a compile failed" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=exports.map -o ../variants/lib$name.so $objs ../../build/var_$name/r4x16_enc_chain_pk.o
echo built ../variants/lib$name.so
