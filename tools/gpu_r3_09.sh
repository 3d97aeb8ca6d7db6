#!/bin/bash
# round 3, call 9: persistent form of the register-cached row sweep (next row's list prefetched through LDS): bits + C4 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_families.py -k "cached or regcached or heavy_tailed or c4" > gpurun_out/r3_09_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_09_pytest.log; tail -4 gpurun_out/r3_09_pytest.log
Q="--steps 10 --warmup 3 --pmc off --no-jref --no-cpu-baseline --no-convergence-run"
for P in 1 0; do for F in 1 2; do
  [ "$P" = 0 ] && [ "$F" = 2 ] && continue
  GLRM_HIP_CACHED_PERSIST=$P GLRM_HIP_CACHED_PERSIST_FILL=$F timeout 300 python bench.py $Q > gpurun_out/r3_09_c4_p${P}f${F}.json 2> gpurun_out/r3_09_c4_p${P}f${F}.err
  python - "$P" "$F" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r3_09_c4_p{sys.argv[1]}f{sys.argv[2]}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("persist", sys.argv[1], "fill", sys.argv[2], "ms/step %.1f row %.2f col %.2f obj %.12g trials %.5f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["objective"]["after_warmup_and_steps"],k["mean_trials_per_row"]))
PY
done; done 2>&1 | tee gpurun_out/r3_09_ab.txt
