mkdir -p gpurun_out/r04q
(time python bench.py) > gpurun_out/r04q/r04_bench_default.json 2> gpurun_out/r04q/bench_default.err
tail -4 gpurun_out/r04q/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04q/r04_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["lane_weighted_frac"])
for c,l in d["other_configs"].items(): print(c, l.get("value"), l.get("ms_per_step"), l["roofline"]["frac"], l["roofline"]["lane_weighted_frac"], l["cpu_baseline"]["value"])
PY
