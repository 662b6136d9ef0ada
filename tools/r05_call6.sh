#!/bin/bash
set -u
O=gpurun_out/r05f; mkdir -p $O
python3 tools/locality_probe.py ibl-lens.toml 2048 2048 1024 $O/locality_c5.json
python3 tools/locality_probe.py mesh-box.toml 1920 1370 1024 $O/locality_c4.json
python3 tools/locality_probe.py cbox-spheres.toml 1024 1024 1024 $O/locality_c2.json
