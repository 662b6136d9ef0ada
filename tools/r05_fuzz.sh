#!/bin/bash
# round 5 fuzz totals on the final build (DESIGN Appendix B)
set -u
O=gpurun_out/r05y; mkdir -p $O
python3 tools/strong_rank_probe.py ibl-lens.toml 2048 2048 8192 0 $O/r05_strong_rank_c5.json 2 2>&1 | tail -4
python3 tools/fuzz_parity.py 500000 1800 48 32 40 > $O/fuzz_default_40spp.txt 2>&1; tail -1 $O/fuzz_default_40spp.txt
LR_BAND_PIX=512 LR_SUB_SHIFT=7 python3 tools/fuzz_parity.py 510000 1500 48 32 16 > $O/fuzz_bands_16spp.txt 2>&1; tail -1 $O/fuzz_bands_16spp.txt
python3 tools/fuzz_parity.py 520000 1000 48 32 8 26 hostile > $O/fuzz_hostile.txt 2>&1; tail -1 $O/fuzz_hostile.txt
python3 tools/fuzz_parity.py 530000 600 48 32 24 120 > $O/fuzz_120obj.txt 2>&1; tail -1 $O/fuzz_120obj.txt
LR_SUB_SHIFT=8 python3 tools/fuzz_parity.py 540000 300 96 64 16 600 > $O/fuzz_600obj.txt 2>&1; tail -1 $O/fuzz_600obj.txt
for f in $O/fuzz_*.txt; do echo $f; grep -c "^seed" $f; grep "worst rel err" $f | sed -E 's/.*worst rel err ([0-9.e+-]+).*/\1/' | sort -g | tail -1; grep -E "ABOVE|ERROR" $f | head -3; done
