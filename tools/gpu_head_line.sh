#!/usr/bin/env bash
# `python bench.py` exactly as the driver runs it (no flags), timed by the shell; the line -> gpurun_out/<tag>_bench_default_head.json
set -u
TAG=${1:-r05}; OUT=gpurun_out; mkdir -p $OUT
t0=$(date +%s.%N)
python bench.py 2> $OUT/${TAG}_head_err.log | tail -1 > $OUT/${TAG}_bench_default_head.json
t1=$(date +%s.%N)
python - <<PY
import json
print("bench.py wall", round($t1 - $t0, 1), "s")
d = json.load(open("$OUT/${TAG}_bench_default_head.json"))
r = d["roofline"]
print(d["ms_per_step"], d["value"], r["frac"], r["traffic"], r.get("traffic_frac"), r.get("mfma_busy"), r.get("traffic_source"))
print({k: v["ms_per_step"] for k, v in d["regimes"].items()})
PY
