#!/bin/bash
# usage (GPU box): tools/pmc_one.sh <tag> <label> <scene> <W> <H> <spp> <flags>    SQ issue counters of one workload -> gpurun_out/<tag>/pmc_<label>.json
# (rocprofv3 gets the python program directly after `--`; counter passes carry no trace options)
TAG=$1; LABEL=$2; SCENE=$3; W=$4; H=$5; SPP=$6; FLAGS=${7:-1}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ1 -d $OUT/p1_$LABEL -o p -- python3 tools/quick_perf.py $SCENE $W $H $SPP 0 $FLAGS > $OUT/pmc1_$LABEL.log 2>&1
rocprofv3 --pmc $SQ2 -d $OUT/p2_$LABEL -o p -- python3 tools/quick_perf.py $SCENE $W $H $SPP 0 $FLAGS > $OUT/pmc2_$LABEL.log 2>&1
python3 tools/rocpd_export.py pmc $OUT/pmc_$LABEL.json $OUT/p1_$LABEL/p_results.db $OUT/p2_$LABEL/p_results.db scene=$SCENE width=$W height=$H spp=$SPP flags=$FLAGS when=$(date +%s) > /dev/null 2>&1
rm -rf $OUT/p1_$LABEL $OUT/p2_$LABEL
python3 - <<PY
import json
d=json.load(open("$OUT/pmc_$LABEL.json"))["kernels"]
for k,v in d.items():
    if not (k.startswith("k_path") or k.startswith("k_resident") or k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_shadow")): continue
    g=lambda n: v.get(n,0.0)
    wc=max(g("SQ_WAVE_CYCLES"),1)
    print("$LABEL %-28s us %9.1f VALU %.4e (trans %.2e) SALU %.3e lanes/instr %4.1f parked %.3f issue-stalled %.3f LDS %.2e SMEM %.2e VMEM %.2e clockGHz %.3f"%(k[:28], g("avg_us_in_pmc_pass"), g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU_TRANS_F32"), g("SQ_INSTS_SALU"), g("SQ_THREAD_CYCLES_VALU")/max(g("SQ_INSTS_VALU"),1), g("SQ_WAIT_ANY")/wc, g("SQ_WAIT_INST_ANY")/wc, g("SQ_INSTS_LDS"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_VMEM_RD")+g("SQ_INSTS_VMEM_WR"), g("GRBM_GUI_ACTIVE")/8/max(g("avg_us_in_pmc_pass"),1)/1e3))
PY
