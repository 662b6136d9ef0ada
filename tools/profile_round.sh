#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/profile_round.sh <tag> [configs...]   (default: c2 c3 c3p c3b c4 c5)
# Collects what DESIGN.md section 6 and bench.py's roofline blocks quote, into gpurun_out/<tag>/ (scratch); the exported
# summaries land in gpurun_out/<tag>/profiles/<tag>_* and are copied into profiles/ (tracked) after the call:
#   per config: full-size throughput (tools/quick_perf.py), rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE
#   passes), SQ issue counters (the VALU roof), L2 / L1 counters, and the bench.py JSON line with those attached.
# rocprofv3 gets the python program directly after `--` (no env / bash hop: the profiler has initialised the GPU).
set -u
TAG=${1:-r03}
shift || true
CFGS=${*:-c2 c3 c3p c3b c4 c5}
OUT=$PWD/gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p "$OUT" "$PROF" profiles
REV=$(cat .git_rev 2>/dev/null || echo unknown)           # written before the call: git rev-parse --short HEAD > .git_rev (the GPU box has no .git); `build` below is what ties a profile to a library
BUILD=$(python3 -c "from lumillyrender_amd import device; print(device.build_id())")     # content hash of csrc/* + flags inside the library that is profiled
export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"
TCC="TCC_HIT TCC_MISS TCC_REQ"
TCP="TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"

run_cfg() {   # name scene W H spp(full) spp(profile passes) bench_steps
  local name=$1 scene=$2 W=$3 H=$4 spp=$5 pspp=$6 steps=$7
  echo "== $name full size: $scene ${W}x${H} $spp spp"
  python3 tools/quick_perf.py $scene $W $H $spp > $OUT/${name}_perf.log 2>&1
  tail -1 $OUT/${name}_perf.log | cut -c1-260
  cp $OUT/${name}_perf.log $PROF/${TAG}_${name}_quick_perf.jsonl
  python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_perf_p.log 2>&1
  local slots=$(tail -1 $OUT/${name}_perf_p.log | python3 -c "import json,sys; print(json.loads(sys.stdin.read()).get('path_slots', 0))")
  local wl="config=$name scene=$scene width=$W height=$H spp=$pspp path_slots=$slots rev=$REV build=$BUILD when=$(date +%s)"
  rocprofv3 --kernel-trace --stats -d $OUT/${name}_kt -o kt -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_kt.log 2>&1
  python3 tools/rocpd_export.py stats $OUT/${name}_kt/kt_results.db $PROF/${TAG}_${name}_kernel_stats.csv > $OUT/${name}_kt_stats.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/${name}_fetch -o p -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/${name}_write -o p -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_write.log 2>&1
  python3 tools/rocpd_export.py traffic $OUT/${name}_fetch/p_results.db $OUT/${name}_write/p_results.db $PROF/${TAG}_traffic_${name}.json $wl > /dev/null 2>&1
  rocprofv3 --pmc $SQ1 -d $OUT/${name}_sq1 -o p -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_sq1.log 2>&1
  rocprofv3 --pmc $SQ2 -d $OUT/${name}_sq2 -o p -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_sq2.log 2>&1
  rocprofv3 --pmc $TCC -d $OUT/${name}_tcc -o p -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_tcc.log 2>&1
  rocprofv3 --pmc $TCP -d $OUT/${name}_tcp -o p -- python3 tools/quick_perf.py $scene $W $H $pspp > $OUT/${name}_tcp.log 2>&1
  local dbs=""; for p in sq1 sq2 tcc tcp; do [ -f $OUT/${name}_$p/p_results.db ] && dbs="$dbs $OUT/${name}_$p/p_results.db"; done
  python3 tools/rocpd_export.py pmc $PROF/${TAG}_pmc_${name}.json $dbs $wl > $OUT/${name}_pmc_export.log 2>&1
  rm -rf $OUT/${name}_kt $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq1 $OUT/${name}_sq2 $OUT/${name}_tcc $OUT/${name}_tcp   # raw databases are large
  cp $PROF/${TAG}_traffic_${name}.json $PROF/${TAG}_pmc_${name}.json profiles/ 2>/dev/null     # bench.py attaches them when the workload matches
  python3 bench.py --config $name --steps $steps --warmup 1 > $PROF/${TAG}_bench_${name}.json 2> $OUT/${name}_bench.err
  cut -c1-400 $PROF/${TAG}_bench_${name}.json
}

for c in $CFGS; do
  case $c in
    c2) run_cfg c2 cbox-spheres.toml 1024 1024 1024 1024 10 ;;
    c3) run_cfg c3 brdf-row.toml 960 540 4096 4096 10 ;;
    c3p) run_cfg c3p brdf-row-phong.toml 960 540 4096 4096 10 ;;
    c3b) run_cfg c3b brdf-row-blinn-phong.toml 960 540 4096 4096 10 ;;
    c4) run_cfg c4 mesh-box.toml 1920 1370 2048 2048 3 ;;
    c5) run_cfg c5 ibl-lens.toml 2048 2048 8192 8192 2 ;;
  esac
done
du -sh $OUT
