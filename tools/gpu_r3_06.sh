#!/bin/bash
# round 3, call 6: general sweeps with register-resident blocks (d <= 8) and the in-kernel exp / log / reciprocal: parity + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q tests/test_gpu_multidim.py tests/test_gpu_fuzz.py tests/test_reference_scripts.py tests/test_gpu_impute.py tests/test_multi_in_process.py tests/test_gpu_crossval.py > gpurun_out/r3_06_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_06_pytest.log; tail -6 gpurun_out/r3_06_pytest.log
for MIX in mnl ordinal mixed; do
  for R in 1 0; do
    echo "== $MIX GLRM_HIP_MULTI_REGS=$R"
    GLRM_HIP_MULTI_REGS=$R timeout 300 python tests/perf/bench_multi.py --mix $MIX --iters 6 2>&1 | grep "hip:"
  done
done 2>&1 | tee gpurun_out/r3_06_multi_ab.txt
echo "== mnl, round 2 formulas + LDS blocks (libglrm_hip_libm.so)" | tee -a gpurun_out/r3_06_multi_ab.txt
GLRM_HIP_LIB_PATH=$PWD/lowrankmodels.jl_amd/libglrm_hip_libm.so GLRM_HIP_MULTI_REGS=0 timeout 300 python tests/perf/bench_multi.py --mix mnl --iters 6 2>&1 | grep "hip:" | tee -a gpurun_out/r3_06_multi_ab.txt
# the default bench line (C4) with every leg: the command the driver runs (call 5 lost it to a missing /usr/bin/time)
S=$(date +%s); timeout 900 python bench.py > gpurun_out/r3_06_bench_c4.json 2> gpurun_out/r3_06_bench_c4.err; echo "bench C4 exit $? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r3_06_bench_c4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_06_bench_c4.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("C4 ms/step %.1f row %.2f col %.2f value %.4g frac %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"]))
print(json.dumps(d["kernels"]["row_sweep"]), json.dumps(d["step_model"]))
print(json.dumps(d["to_ref_objective"])); print(json.dumps(d["cpu_baseline"]))
PY
