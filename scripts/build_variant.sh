#!/bin/bash
# Build a variant of libhosrender.so with extra -D flags for timing experiments:  scripts/build_variant.sh NAME -DFOO=1 ...
# Use it with HOS_LIB_PATH=build/variants/NAME/libhosrender.so; VARIANT_SRC=<file.hip> names the translation unit the flags are for
# (default hos_gemmp.hip; the other objects are built once per variant directory)
set -e
NAME=$1; shift
D=build/variants/$NAME; mkdir -p $D
for f in hosnerf_amd/csrc/*.hip; do
  o=$D/$(basename ${f%.hip}).o
  if [ "$(basename $f)" = "${VARIANT_SRC:-hos_gemmp.hip}" ] || [ ! -f $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -Ihosnerf_amd/csrc -Wno-unused-result "$@" -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libhosrender.so $D/*.o
echo built $D/libhosrender.so
