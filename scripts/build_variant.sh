#!/bin/bash
# Developer A/B harness: builds librtcore with extra -D flags into variants/<name>.so (git-ignored); select with RTCORE_LIB.
# usage: scripts/build_variant.sh <name> [-DFLAG ...]
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
mkdir -p $ROOT/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-parameter \
  -I$ROOT/include -I$ROOT/raytracer_amd/csrc "$@" -shared -o $ROOT/variants/$NAME.so $ROOT/raytracer_amd/csrc/rtcore.hip
echo built variants/$NAME.so
