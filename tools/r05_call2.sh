#!/bin/bash
# round 5, GPU call 2: wave timelines (ramp / tail) of the four configs, whole frame and a 1/8 share; tile edge at world 1;
# all-ranks probe with the diagonal deal at 32 and 16 px
set -u
O=gpurun_out/r05b; mkdir -p $O
export LR_HIP_LIB=$PWD/build/v_timeline.so
for a in "cbox-spheres.toml 1024 1024 1024" "brdf-row.toml 960 540 4096" "mesh-box.toml 1920 1370 2048" "ibl-lens.toml 2048 2048 2048"; do
  set -- $a
  echo "== $a world 1" >> $O/timeline.txt; python3 tools/timeline_probe.py $a 1 0 >> $O/timeline.txt 2>&1
  echo "== $a world 8 rank 3" >> $O/timeline.txt; python3 tools/timeline_probe.py $a 8 3 >> $O/timeline.txt 2>&1
done
unset LR_HIP_LIB
cat $O/timeline.txt
python3 tools/tile_size_probe.py cbox-spheres.toml 1024 1024 1024 $O/tile_size_c2.json
python3 tools/tile_size_probe.py brdf-row.toml 960 540 4096 $O/tile_size_c3.json
python3 tools/tile_size_probe.py mesh-box.toml 1920 1370 1024 $O/tile_size_c4.json
python3 tools/tile_size_probe.py ibl-lens.toml 2048 2048 1024 $O/tile_size_c5.json
for t in 32 16; do
python3 tools/strong_rank_probe.py cbox-spheres.toml 1024 1024 1024 $t $O/strong_diag${t}_c2.json 3 2>&1 | tail -1
python3 tools/strong_rank_probe.py brdf-row.toml 960 540 4096 $t $O/strong_diag${t}_c3.json 3 2>&1 | tail -1
python3 tools/strong_rank_probe.py mesh-box.toml 1920 1370 2048 $t $O/strong_diag${t}_c4.json 2 2>&1 | tail -1
python3 tools/strong_rank_probe.py ibl-lens.toml 2048 2048 2048 $t $O/strong_diag${t}_c5.json 2 2>&1 | tail -1
done
