#!/bin/bash
# usage: [FLAT_EXTRA="..."] [TREE_EXTRA="..."] tools/build_variant.sh <name> [-DFLAG=..]...   ->  build/v_<name>.so
# A build of liblumilly_hip.so with extra defines / compiler switches, for tools/ab4.py.  The three translation units are compiled as
# csrc/Makefile compiles them: "$@" goes to all of them, TREE_EXTRA to lumilly_hip.hip only, FLAT_EXTRA to lr_flat.hip only
# (default: the Makefile's FLATFLAGS; FLAT_EXTRA=" " builds the flat kernels without any switch).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
O="$ROOT/build/obj_$name"; mkdir -p "$O"
cd "$ROOT/lumillyrender_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function"
FLAT=${FLAT_EXTRA--mllvm -enable-post-misched=false}
( /opt/rocm/bin/hipcc $F "$@" $TREE_EXTRA -c -o "$O/main.o" lumilly_hip.hip 2>&1 | grep -E "error" || true ) &
( /opt/rocm/bin/hipcc $F "$@" $FLAT -c -o "$O/flat.o" lr_flat.hip 2>&1 | grep -E "error" || true ) &
( /opt/rocm/bin/hipcc $F "$@" -c -o "$O/lbvh.o" lr_lbvh.hip 2>&1 | grep -E "error" || true ) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/build/v_$name.so" "$O/main.o" "$O/flat.o" "$O/lbvh.o"
rm -rf "$O"
ls -la "$ROOT/build/v_$name.so"
