#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
run() { name=$1; shift; timeout 900 python bench.py --steps 5 --warmup 3 "$@" > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/cfg_$name.json")); k=d["kernels"]; c=d["config"]
    print("$name: %.3g upd/s ms/step %.2f row %.2f (%s) col %.2f (%s) trials %.3f/%.3f obj %.8g gen %.1fs" % (d["value"], d["ms_per_step"], k["row_sweep_ms"], c["row_sweep"], k["col_sweep_ms"], c["col_sweep"], k["mean_trials_per_row"], k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"], d["setup_s"]["generate"]))
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/cfg_$name.err").read()[-800:])
PY
}
run C4_full --config C4 --rows-per-gpu 10000000
run C4_full_tiled --config C4 --rows-per-gpu 10000000 --tiled 2
run C4_shard --config C4 --rows-per-gpu 1250000
run C5_1M --config C5 --rows-per-gpu 1000000
run C5_1M_gather --config C5 --rows-per-gpu 1000000 --tiled 1
