#!/bin/bash
# round 3, call 4: LogisticLoss in the reference's own rounding structure (one exponential): fuzz + loss parity, C5-family A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_families.py tests/test_gpu_multidim.py tests/test_gpu_crossval.py tests/test_reference_scripts.py tests/test_gpu_impute.py tests/test_reference_notebook.py > gpurun_out/r3_04_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_04_pytest.log; tail -6 gpurun_out/r3_04_pytest.log
Q="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
for L in libglrm_hip.so libglrm_hip_libm.so libglrm_hip.so; do
  timeout 300 python tests/perf/ab_lib.py $L $Q > gpurun_out/r3_04_c5fam_$L.json 2> gpurun_out/r3_04_c5fam_$L.err
  python - "$L" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r3_04_c5fam_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1], "ms/step %.1f row %.2f col %.2f trials %.3f %.3f obj %.10g" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],k["mean_trials_per_row"],k["mean_trials_per_col"], d["objective"]["after_warmup_and_steps"]))
PY
done 2>&1 | tee gpurun_out/r3_04_ab.txt
for C in SQ_INSTS_VALU WRITE_SIZE; do
  rm -rf /tmp/pmc_$C; ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py $Q --steps 2 > /dev/null 2> /tmp/pmc_$C.err )
  python - "$C" <<'PY'
import csv,glob,sys,re,collections
C=sys.argv[1]
acc=collections.defaultdict(lambda:[0.0,0])
for p in glob.glob(f"/tmp/pmc_{C}/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        if r.get("Counter_Name")==C and "tiled" in r["Kernel_Name"]:
            k=re.sub(r"\(.*","",r["Kernel_Name"])[:70]
            acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
for k,(v,n) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:4]:
    print(f"PMC product {C} {k} total={v:.4g} dispatches={n} mean={v/n:.4g}")
PY
done 2>&1 | tee -a gpurun_out/r3_04_ab.txt
