#!/bin/bash
# round 3, call 12: final state -- whole -m gpu suite, smoke, the default bench line (what the driver runs), shard geometry N = 2 / 4 / 8
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; T=r3_12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${T}_smoke.log
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log; grep -E "FAILED|ERROR" gpurun_out/${T}_pytest.log | head
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_c4.json 2> gpurun_out/${T}_bench_c4.err; echo "bench C4 exit $? wall $(( $(date +%s) - S )) s"
rm -rf gpurun_out/${T}_prof_c4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_c4 -o bench -- python $R/bench.py --steps 10 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/${T}_prof_c4_bench.json 2> $R/gpurun_out/${T}_prof_c4.err); echo "prof C4 exit $?"
for N in 2 4 8; do
  timeout 300 python bench.py --emulate-rank 0 --of $N --steps 10 --warmup 3 > gpurun_out/${T}_shard_${N}.json 2> gpurun_out/${T}_shard_${N}.err; echo "shard $N exit $?"
done
find gpurun_out -name "*kernel_trace*" -size +4M -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_12_bench_*.json")+glob.glob("gpurun_out/r3_12_shard_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "mode" in d: print(f, d["measured_ms"], d["predicted_iteration_ms"]); continue
        k=d["kernels"]; print(f, "ms/step %.1f row %.2f col %.2f value %.4g frac %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"]), d["kernels"]["row_sweep"], d["step_model"]["GBps"], d["to_ref_objective"]["gpu_seconds_to_J_ref"], d["cpu_baseline"]["value"], d["roofline"]["traffic"])
    except Exception as e: print(f,"ERR",e)
PY
