#!/bin/bash
# round 3, call 2: in-kernel fp64 exp / log1p (one exponential per LogisticLoss observation) -- parity, same-box A/B against round 2's
# formulas (libglrm_hip_libm.so) on the C5-family shape with VALU / scratch counters; lists borrowed in place; C5 at its stated size.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_families.py tests/test_gpu_multidim.py tests/test_gpu_crossval.py tests/test_reference_scripts.py tests/test_gpu_impute.py > gpurun_out/r3_02_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_02_pytest.log; tail -4 gpurun_out/r3_02_pytest.log
Q="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
for L in libglrm_hip.so libglrm_hip_libm.so; do
  timeout 300 python tests/perf/ab_lib.py $L $Q > gpurun_out/r3_02_c5fam_$L.json 2> gpurun_out/r3_02_c5fam_$L.err; echo "c5fam $L exit $?"
  for C in SQ_INSTS_VALU WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/pmc_$C; ( cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/tests/perf/ab_lib.py $L $Q --steps 2 > /dev/null 2> /tmp/pmc_$C.err )
    python - "$L" "$C" <<'PY'
import csv,glob,sys,re,collections
L,C=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda:[0.0,0])
for p in glob.glob(f"/tmp/pmc_{C}/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        if r.get("Counter_Name")==C:
            k=re.sub(r"\(.*","",r["Kernel_Name"])[:70]
            acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
for k,(v,n) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:6]:
    print(f"PMC {L} {C} {k} total={v:.4g} dispatches={n} mean={v/n:.4g}")
PY
  done
done 2>&1 | tee gpurun_out/r3_02_ab.txt
timeout 900 python bench.py --config C5 --steps 10 --warmup 3 > gpurun_out/r3_02_bench_c5_full.json 2> gpurun_out/r3_02_bench_c5_full.err; echo "C5 full exit $?"
tail -c 600 gpurun_out/r3_02_bench_c5_full.err
python - <<'PY'
import json
for f in ("gpurun_out/r3_02_c5fam_libglrm_hip.so.json","gpurun_out/r3_02_c5fam_libglrm_hip_libm.so.json","gpurun_out/r3_02_bench_c5_full.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
        print(f, "ms/step %.1f row %.2f col %.2f trials %.3f %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],k["mean_trials_per_row"],k["mean_trials_per_col"]), d["config"].get("full_size"), d.get("setup_s"))
    except Exception as e: print(f,"ERR",e)
PY
