#!/bin/bash
# developer helper: a VARIANT build of the library for A-B-A-B runs (tools/abab_libs.sh) -- the named translation units are recompiled with extra flags, every other
# object comes from the product build (ndrustfft_amd/csrc/_build).   usage: tools/build_variant.sh <name> "<extra flags>" <file.hip> [...]   ->  tools/_ab/<name>.so
set -e
NAME=$1; FLAGS=$2; shift 2
C=$(dirname $0)/../ndrustfft_amd/csrc; B=$C/_build_var_$NAME; mkdir -p $B $(dirname $0)/_ab
OBJS=""
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$C/_build -I$C --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-gpu-rdc -munsafe-fp-atomics $FLAGS -c $C/$f -o $B/${f%.hip}.o &
done
wait
for o in $C/_build/*.o; do b=$(basename $o); if [ -f $B/$b ]; then OBJS="$OBJS $B/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $(dirname $0)/_ab/$NAME.so $OBJS -ldl -lpthread
echo "built tools/_ab/$NAME.so"
