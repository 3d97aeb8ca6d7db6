#!/bin/bash
# round 3, call 33: bench lines of the LDS-tiled configurations after the DMA staging: C2 (full legs), C5-family, C5 at its stated size + kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; T=r3_33
timeout 500 python bench.py --config C2 --steps 20 --warmup 5 > gpurun_out/${T}_bench_c2.json 2> gpurun_out/${T}_bench_c2.err; echo "bench C2 exit $?"
timeout 300 python bench.py --config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 6 --warmup 2 > gpurun_out/${T}_bench_c5fam.json 2> gpurun_out/${T}_bench_c5fam.err; echo "bench C5-family exit $?"
timeout 700 python bench.py --config C5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_c5_full.json 2> gpurun_out/${T}_bench_c5_full.err; echo "bench C5 full exit $?"; tail -2 gpurun_out/${T}_bench_c5_full.err
rm -rf gpurun_out/${T}_prof_c5
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_c5 -o bench -- python $R/bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-jref --no-convergence-run --pmc off > $R/gpurun_out/${T}_prof_c5_bench.json 2> $R/gpurun_out/${T}_prof_c5.err); echo "prof C5 exit $?"
find gpurun_out -name "*kernel_trace*" -size +4M -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_33_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
        print(f, "ms/step %.2f row %.2f col %.2f value %.4g frac %.3f trials %.3f %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"],k["mean_trials_per_row"],k["mean_trials_per_col"]), d.get("convergence"))
    except Exception as e: print(f,"ERR",e)
PY
