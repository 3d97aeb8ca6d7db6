#!/bin/bash
# round 3, call 1: whole GPU suite after the signature / class-plan refactor + new family tests, C4 bench sanity, shard geometry N=8
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r3_01_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_01_pytest.log
tail -5 gpurun_out/r3_01_pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-jref --no-cpu-baseline --pmc off --no-convergence-run > gpurun_out/r3_01_bench_c4.json 2> gpurun_out/r3_01_bench_c4.err
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3_01_bench_c4.json"))
    print("C4", d["ms_per_step"], d["kernels"]["row_sweep_ms"], d["kernels"]["col_sweep_ms"], d["config"]["row_sweep"], d["config"]["col_sweep"])
except Exception as e: print("bench failed", e)
PY
for N in 8 2; do
timeout 300 python bench.py --emulate-rank 0 --of $N --steps 10 --warmup 3 > gpurun_out/r3_01_shard_${N}.json 2> gpurun_out/r3_01_shard_${N}.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3_01_shard_${N}.json"))
    print("shard $N", d["measured_ms"], d["families"], d["predicted_iteration_ms"])
except Exception as e: print("shard failed", e)
PY
done
