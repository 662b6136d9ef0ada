#!/bin/bash
# usage (GPU box): tools/ab_libs.sh "<scene W H spp>" variant[,variant] lib1.so lib2.so ...   ("" = the product library)
# one process per library build (LR_HIP_LIB), interleaved rounds inside each process (tools/ab_pipelines.py)
WL=$1; VAR=$2; shift 2
for lib in "$@"; do
  echo "== ${lib:-product}"
  if [ -n "$lib" ]; then LR_HIP_LIB=$PWD/$lib timeout ${AB_TIMEOUT:-600} python3 tools/ab_pipelines.py $WL 2 $VAR; else timeout ${AB_TIMEOUT:-600} python3 tools/ab_pipelines.py $WL 2 $VAR; fi
done
