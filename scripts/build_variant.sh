#!/bin/bash
# Developer A/B harness: builds librtcore with extra -D flags into variants/<name>.so (git-ignored); select with RTCORE_LIB.
# usage: scripts/build_variant.sh <name> [-DFLAG ...]
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
mkdir -p $ROOT/variants
make -s -C $ROOT/raytracer_amd/csrc $ROOT/raytracer_amd/csrc/build/rt_sort.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-parameter \
  -I$ROOT/include -I$ROOT/raytracer_amd/csrc "$@" -c -o $ROOT/variants/$NAME.o $ROOT/raytracer_amd/csrc/rtcore.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/variants/$NAME.so $ROOT/variants/$NAME.o $ROOT/raytracer_amd/csrc/build/rt_sort.o -ldl
rm -f $ROOT/variants/$NAME.o
echo built variants/$NAME.so
