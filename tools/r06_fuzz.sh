#!/bin/bash
# round 6 fuzz totals on the final build (DESIGN Appendix B): random scenes, device vs oracle, four pipelines per seed
set -u
cd $GRAFT_REPO_ROOT
M=${FUZZ_MULT:-1}                    # times as many seeds per class
F=${FUZZ_OFFSET:-0}                  # added to every first seed: a second run covers other scenes
O=gpurun_out/r06_fuzz$( [ $F = 0 ] || echo _$F ); mkdir -p $O
timeout $((1500 * M)) python3 tools/fuzz_parity.py $((600000 + F)) $((1200 * M)) 48 32 40 > $O/fuzz_default_40spp.txt 2>&1; tail -1 $O/fuzz_default_40spp.txt
timeout $((900 * M)) python3 tools/fuzz_parity.py $((620000 + F)) $((800 * M)) 48 32 8 26 hostile > $O/fuzz_hostile.txt 2>&1; tail -1 $O/fuzz_hostile.txt
timeout $((900 * M)) python3 tools/fuzz_parity.py $((630000 + F)) $((500 * M)) 48 32 24 120 > $O/fuzz_120obj.txt 2>&1; tail -1 $O/fuzz_120obj.txt
timeout $((900 * M)) python3 tools/fuzz_parity.py $((640000 + F)) $((250 * M)) 96 64 16 600 > $O/fuzz_600obj.txt 2>&1; tail -1 $O/fuzz_600obj.txt
export LR_HIP_LIB=$PWD/lumillyrender_amd/liblumilly_hip_knobs.so LR_BAND_PIX=512 LR_SUB_SHIFT=7
timeout $((900 * M)) python3 tools/fuzz_parity.py $((610000 + F)) $((800 * M)) 48 32 16 > $O/fuzz_bands_16spp.txt 2>&1; tail -1 $O/fuzz_bands_16spp.txt
unset LR_HIP_LIB LR_BAND_PIX LR_SUB_SHIFT
for f in $O/fuzz_*.txt; do echo $f; grep -c "^seed" $f; grep "worst rel err" $f | sed -E 's/.*worst rel err ([0-9.e+-]+).*/\1/' | sort -g | tail -1; grep -E "ABOVE|ERROR" $f | head -3; done
