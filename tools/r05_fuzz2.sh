#!/bin/bash
# more fuzz on the final build of round 5 (culling slack K = 8)
set -u
O=gpurun_out/r05x; mkdir -p $O
python3 tools/fuzz_traversal.py 4000 400 > $O/fuzz_trav.txt 2>&1; grep -v "outside own box 0$" $O/fuzz_trav.txt | tail -8
python3 tools/fuzz_parity.py 600000 4000 48 32 40 > $O/fuzz_default_40spp.txt 2>&1; tail -1 $O/fuzz_default_40spp.txt
LR_BAND_PIX=512 LR_SUB_SHIFT=7 python3 tools/fuzz_parity.py 610000 2500 48 32 16 120 > $O/fuzz_bands_120obj.txt 2>&1; tail -1 $O/fuzz_bands_120obj.txt
python3 tools/fuzz_parity.py 620000 2000 48 32 8 120 hostile > $O/fuzz_hostile_120.txt 2>&1; tail -1 $O/fuzz_hostile_120.txt
for f in $O/fuzz_*spp.txt $O/fuzz_bands_120obj.txt $O/fuzz_hostile_120.txt; do echo $f; grep -c "^seed" $f; grep "worst rel err" $f | sed -E 's/.*worst rel err ([0-9.e+-]+).*/\1/' | sort -g | tail -1; grep -E "ABOVE|ERROR" $f | head -3; done
