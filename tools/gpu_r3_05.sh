#!/bin/bash
# round 3, call 5: the bench lines of this round with every leg (C4 default incl. fixture J_ref + two-sample cpu_baseline + step model),
# rocprofv3 kernel stats of C4 and of C5 at its stated size, shard geometry N = 4 / 8 with TCC hit rates, C2 / C3 regression lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
T=r3_05
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${T}_smoke.log
/usr/bin/time -f "bench wall %e s" timeout 900 python bench.py > gpurun_out/${T}_bench_c4.json 2> gpurun_out/${T}_bench_c4.err; echo "bench C4 exit $?"; tail -2 gpurun_out/${T}_bench_c4.err
rm -rf gpurun_out/${T}_prof_c4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_c4 -o bench -- python $R/bench.py --steps 10 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/${T}_prof_c4_bench.json 2> $R/gpurun_out/${T}_prof_c4.err); echo "prof C4 exit $?"
for N in 4 8; do
  timeout 300 python bench.py --emulate-rank 0 --of $N --steps 10 --warmup 3 > gpurun_out/${T}_shard_${N}.json 2> gpurun_out/${T}_shard_${N}.err; echo "shard $N exit $?"
done
# TCC hit rate + fetched bytes of one rank's kernels at N = 8 (separate PMC passes)
for C in "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
  D=/tmp/pmc_shard_$(echo $C | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -o pmc -- python $R/bench.py --emulate-rank 0 --of 8 --steps 2 --warmup 1 > /dev/null 2> $D.err)
  python - "$D" <<'PY'
import csv,glob,sys,re,collections
acc=collections.defaultdict(lambda:collections.defaultdict(float)); cnt=collections.Counter()
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        k=re.sub(r"\(.*","",r["Kernel_Name"])[:60]
        if "sweep" in k or "pass" in k:
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    print("PMC shard8", k, {c:(x, cnt[(k,c)]) for c,x in v.items()}, "hit rate %.3f" % (v["TCC_HIT_sum"]/(v["TCC_HIT_sum"]+v["TCC_MISS_sum"])) if "TCC_HIT_sum" in v else "")
PY
done 2>&1 | tee gpurun_out/${T}_shard8_pmc.txt
timeout 600 python bench.py --config C2 > gpurun_out/${T}_bench_c2.json 2> gpurun_out/${T}_bench_c2.err; echo "bench C2 exit $?"
timeout 900 python bench.py --config C3 --steps 10 --quad-gram --no-convergence-run > gpurun_out/${T}_bench_c3gram.json 2> gpurun_out/${T}_bench_c3gram.err; echo "bench C3 quad_gram exit $?"
timeout 900 python bench.py --config C5 --steps 10 --warmup 3 > gpurun_out/${T}_bench_c5.json 2> gpurun_out/${T}_bench_c5.err; echo "bench C5 full exit $?"; tail -2 gpurun_out/${T}_bench_c5.err
rm -rf gpurun_out/${T}_prof_c5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_c5 -o bench -- python $R/bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/${T}_prof_c5_bench.json 2> $R/gpurun_out/${T}_prof_c5.err); echo "prof C5 exit $?"
find gpurun_out -name "*kernel_trace*" -size +4M -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_05_bench_*.json")+glob.glob("gpurun_out/r3_05_shard_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if "mode" in d: print(f, d["measured_ms"], d["predicted_iteration_ms"]); continue
        k=d["kernels"]; print(f, "ms/step %.1f row %.2f col %.2f value %.4g frac %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"]), d.get("step_model") and d["step_model"]["GBps"], d.get("to_ref_objective",{}).get("gpu_first_iteration_at_or_below_J_ref"), d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f,"ERR",e)
PY
