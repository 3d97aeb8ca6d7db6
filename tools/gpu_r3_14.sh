#!/bin/bash
# round 3, call 14: one-wave workgroups + half-residency slices as the default of the phase-aligned passes: parity subset, grid of the
# persistent row sweep, default bench, shard geometry
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3_14
timeout 1200 python -m pytest -m gpu -q tests/test_gpu_families.py tests/test_gpu_multirank.py tests/test_gpu_parity.py tests/test_gpu_bench_modes.py tests/test_gpu_fullsize.py -k "not c2_full and not c3_full and not c5_full" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log
Q="--steps 8 --warmup 2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for F in 100 75 50 150; do
  GLRM_HIP_CACHED_PERSIST_FILL=$F timeout 300 python bench.py $Q > gpurun_out/${T}_tmp.json 2> gpurun_out/${T}_tmp.err
  python - "$F" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r3_14_tmp.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("persist grid", sys.argv[1], "% ms/step %.1f row %.2f col %.2f obj %.12g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/${T}_persist_grid.txt
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_c4.json 2> gpurun_out/${T}_bench_c4.err; echo "bench C4 exit $? wall $(( $(date +%s) - S )) s"
for N in 2 4 8; do
  timeout 300 python bench.py --emulate-rank 0 --of $N --steps 10 --warmup 3 > gpurun_out/${T}_shard_${N}.json 2> gpurun_out/${T}_shard_${N}.err; echo "shard $N exit $?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_14_bench_*.json")+glob.glob("gpurun_out/r3_14_shard_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "mode" in d: print(f, d["measured_ms"], d["predicted_iteration_ms"]); continue
        k=d["kernels"]; print(f, "ms/step %.1f row %.2f col %.2f value %.4g frac %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"]), d["kernels"]["row_sweep"]["frac"], d["step_model"]["GBps"], d["to_ref_objective"]["gpu_seconds_to_J_ref"], d["cpu_baseline"]["value"], d["roofline"]["traffic"])
    except Exception as e: print(f,"ERR",e)
PY
