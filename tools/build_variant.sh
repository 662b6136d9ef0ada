#!/bin/bash
# usage: [FLAT_EXTRA="..."] [TREE_EXTRA="..."] tools/build_variant.sh <name> [-DFLAG=..]...   ->  build/v_<name>.so
# A build of liblumilly_hip.so with extra defines / compiler switches, for tools/ab4.py.  The three translation units are compiled as
# csrc/Makefile compiles them: "$@" goes to all of them, TREE_EXTRA to lumilly_hip.hip only, FLAT_EXTRA to lr_flat.hip only
# (default: the Makefile's FLATFLAGS; FLAT_EXTRA=" " builds the flat kernels without any switch).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
O="$ROOT/build/obj_$name"; rm -rf "$O"; mkdir -p "$O"          # never link objects an aborted run left behind
cd "$ROOT/lumillyrender_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-function"
FLAT=${FLAT_EXTRA--mllvm -enable-post-misched=false}
# the three compiles run side by side; each one's exit status is collected (a failed compile aborts: no stale or missing object is linked)
/opt/rocm/bin/hipcc $F "$@" $TREE_EXTRA -c -o "$O/main.o" lumilly_hip.hip > "$O/main.log" 2>&1 & p1=$!
/opt/rocm/bin/hipcc $F "$@" $FLAT -c -o "$O/flat.o" lr_flat.hip > "$O/flat.log" 2>&1 & p2=$!
/opt/rocm/bin/hipcc $F "$@" -c -o "$O/lbvh.o" lr_lbvh.hip > "$O/lbvh.log" 2>&1 & p3=$!
rc=0
wait $p1 || rc=1; wait $p2 || rc=1; wait $p3 || rc=1
grep -h -E "error" "$O"/*.log || true
if [ $rc -ne 0 ]; then echo "build_variant: a compile failed (logs in $O)" >&2; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/build/v_$name.so" "$O/main.o" "$O/flat.o" "$O/lbvh.o"
rm -rf "$O"
ls -la "$ROOT/build/v_$name.so"
