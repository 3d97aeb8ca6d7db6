#!/bin/bash
# Phase-aligned gather passes (glrm_blocked.hip): parity with the family forced on, then A/B at C4 and L2 hit rates.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
GLRM_HIP_BLOCKED=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_multi_in_process.py tests/test_gpu_crossval.py -m gpu -q -x --timeout 600 > gpurun_out/pytest_blocked.log 2>&1; echo "== pytest (blocked forced): $(tail -1 gpurun_out/pytest_blocked.log)"
grep -E "FAILED|ERROR|assert|Error" gpurun_out/pytest_blocked.log | head -10
Q="--config C4 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 3"
GLRM_HIP_BLOCKED=0 timeout 600 python bench.py $Q > gpurun_out/c4_gather.json 2> gpurun_out/c4_gather.err; echo "gather exit $?"
timeout 600 python bench.py $Q > gpurun_out/c4_blocked.json 2> gpurun_out/c4_blocked.err; echo "blocked exit $?"; tail -3 gpurun_out/c4_blocked.err
GLRM_HIP_BLOCKED=1 timeout 600 python bench.py $Q > gpurun_out/c4_blocked_rows.json 2> gpurun_out/c4_blocked_rows.err; echo "blocked rows exit $?"
GLRM_HIP_BLOCKED=2 timeout 600 python bench.py $Q > gpurun_out/c4_blocked_cols.json 2> gpurun_out/c4_blocked_cols.err; echo "blocked cols exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c4_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "ms/iter %.2f" % d["ms_per_step"], "row %.2f col %.2f" % (d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"]), d["config"]["row_sweep"], d["config"]["col_sweep"], "trials %.3f %.3f" % (d["kernels"]["mean_trials_per_row"], d["kernels"]["mean_trials_per_col"]), "obj", d["objective"]["after_warmup_and_steps"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_blk -o pmc -- python $R/bench.py --config C4 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 2 --warmup 2 > $R/gpurun_out/pmc_blk.json 2> $R/gpurun_out/pmc_blk.err); echo "pmc exit $?"
python - <<'PY'
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for p in glob.glob("gpurun_out/pmc_blk/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        per[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k]+=1
for k,cs in per.items():
    if "TCC_HIT_sum" in cs and cs["TCC_HIT_sum"]+cs["TCC_MISS_sum"]>1e8:
        print(k[:90], "dispatches", cnt[k]//2, "hit rate %.3f" % (cs["TCC_HIT_sum"]/(cs["TCC_HIT_sum"]+cs["TCC_MISS_sum"])), "misses %.3g" % cs["TCC_MISS_sum"])
PY
find gpurun_out -name "*kernel_trace*" -size +8M -delete; find gpurun_out -name "*counter_collection.csv" -size +8M -delete
