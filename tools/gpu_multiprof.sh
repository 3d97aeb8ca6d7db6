#!/bin/bash
# Kernel stats + SQ counters of the general sweeps (multi-dimensional losses) on the categorical benchmark.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
ARGS="--m 200000 --n 100 --K 5 --k 10 --iters 3 --mix ${1:-mnl}"
rm -rf gpurun_out/mprof gpurun_out/pmc_m*
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mprof -o multi -- python $R/tests/perf/bench_multi.py $ARGS > $R/gpurun_out/mprof_bench.log 2>&1)
tail -1 gpurun_out/mprof_bench.log
find gpurun_out/mprof -name "*kernel_stats.csv" | head -1 | xargs head -6
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-20)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_m$tag -o pmc -- python $R/tests/perf/bench_multi.py $ARGS > /dev/null 2>&1); echo "pmc $tag exit $?"
done
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/pmc_m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "multi_sweep" in r["Kernel_Name"]:
            key = "rows" if "Lb1" in r["Kernel_Name"] or "<true" in r["Kernel_Name"] else "cols"
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in rows:
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(rows[k].items())})
PY
find gpurun_out -name "*kernel_trace*" -size +8M -delete
