#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG=..]...   ->  build/v_<name>.so  (a build of liblumilly_hip.so with extra defines, for tools/ab4.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/build"
cd "$ROOT/lumillyrender_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -shared -o "$ROOT/build/v_$name.so" lumilly_hip.hip lr_lbvh.hip lr_flat.hip 2>&1 | grep -E "error" || true
ls -la "$ROOT/build/v_$name.so"
