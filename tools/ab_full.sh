#!/bin/bash
# usage (GPU box): tools/ab_full.sh lib1.so lib2.so ...   ("" = the product library): C2 / C4 (1024 spp) / C5 (512 spp) through the default pipeline
for wl in "cbox-spheres.toml 1024 1024 1024" "mesh-box.toml 1920 1370 1024" "ibl-lens.toml 2048 2048 512"; do
  tools/ab_libs.sh "$wl" auto "$@" | grep -E "^==|Msamples" | sed -e 's/"variant": "auto", //' -e 's/"W".*"samples_ok"/"samples_ok"/'
done
