#!/bin/bash
# round 3, call 10: the whole -m gpu suite on the final kernels, the default bench line (C4) + rocprofv3 kernel stats, shard geometry with
# the persistent row sweep, C2 timing builds (NOSTAGE / NOCOMPUTE), SQ counters of the general sweeps before / after
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; T=r3_10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${T}_smoke.log
timeout 1800 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log; tail -4 gpurun_out/${T}_pytest.log; grep -E "FAILED|ERROR" gpurun_out/${T}_pytest.log | head
S=$(date +%s); timeout 900 python bench.py > gpurun_out/${T}_bench_c4.json 2> gpurun_out/${T}_bench_c4.err; echo "bench C4 exit $? wall $(( $(date +%s) - S )) s"
rm -rf gpurun_out/${T}_prof_c4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_c4 -o bench -- python $R/bench.py --steps 10 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/${T}_prof_c4_bench.json 2> $R/gpurun_out/${T}_prof_c4.err); echo "prof C4 exit $?"
for N in 2 4 8; do
  timeout 300 python bench.py --emulate-rank 0 --of $N --steps 10 --warmup 3 > gpurun_out/${T}_shard_${N}.json 2> gpurun_out/${T}_shard_${N}.err; echo "shard $N exit $?"
done
QC="--config C2 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2"
for L in libglrm_hip.so libglrm_hip_NOSTAGE.so libglrm_hip_NOCOMPUTE.so; do
  timeout 300 python tests/perf/ab_lib.py $L $QC > gpurun_out/${T}_c2_$L.json 2> gpurun_out/${T}_c2_$L.err
  python - "$L" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r3_10_c2_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("C2", sys.argv[1], "row %.2f col %.2f trials %.3f %.3f" % (k["row_sweep_ms"],k["col_sweep_ms"],k["mean_trials_per_row"],k["mean_trials_per_col"]))
PY
done 2>&1 | tee gpurun_out/${T}_c2_timing_builds.txt
ARGS="--m 200000 --n 100 --K 5 --k 10 --iters 3 --mix mnl"
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"
for V in new old; do
  rm -rf /tmp/pmc_multi_$V
  if [ $V = new ]; then E=""; else E="GLRM_HIP_MULTI_REGS=0 GLRM_HIP_MULTI_KINDS=0 GLRM_HIP_LIB_PATH=$R/lowrankmodels.jl_amd/libglrm_hip_libm.so"; fi
  (cd /tmp && env $E timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_multi_$V -o pmc -- python $R/tests/perf/bench_multi.py $ARGS > /dev/null 2>&1)
  python - "$V" <<'PY'
import csv, glob, collections, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"/tmp/pmc_multi_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        kn=r["Kernel_Name"]
        if "multi_sweep" in kn or "multi_colpass" in kn:
            key = "rows" if "multi_sweep_kernel<true" in kn else ("colpass_grad" if "colpass_kernel<true" in kn else "colpass_trial")
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in rows:
    v={c: sum(x)/len(x) for c,x in rows[k].items()}
    print("SQ", sys.argv[1], k, {c: "%.4g" % x for c,x in sorted(v.items())}, "VALU-active %.2f wait-any %.2f" % (v["SQ_ACTIVE_INST_VALU"]*4/ v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else 0, v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else 0))
PY
done 2>&1 | tee gpurun_out/${T}_multi_sq.txt
find gpurun_out -name "*kernel_trace*" -size +4M -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_10_bench_*.json")+glob.glob("gpurun_out/r3_10_shard_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "mode" in d: print(f, d["measured_ms"], d["predicted_iteration_ms"]); continue
        k=d["kernels"]; print(f, "ms/step %.1f row %.2f col %.2f value %.4g frac %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"]), d["kernels"]["row_sweep"], d["step_model"]["GBps"], d["to_ref_objective"]["gpu_seconds_to_J_ref"], d["cpu_baseline"]["value"])
    except Exception as e: print(f,"ERR",e)
PY
