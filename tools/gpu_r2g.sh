#!/bin/bash
# Does super-tile-major dispatch keep the staged tiles in L2?  TCC hit / miss and FETCH_SIZE of the row pass with and without the split.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
Q="--config C2 --rows 1000000 --cols 50000 --obs-per-row 1000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 2 --warmup 2"
for S in 0 1; do
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    tag=s${S}_$(echo $C | tr ' ' '_' | cut -c1-12)
    rm -rf gpurun_out/pmc_$tag
    (cd /tmp && GLRM_HIP_ROW_SPLIT=$S timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py $Q > $R/gpurun_out/pmc_$tag.json 2> $R/gpurun_out/pmc_$tag.err); echo "pmc $tag exit $?"
  done
done
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob("gpurun_out/pmc_s*/")):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d+"pmc_counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            if "tiled" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in per.items():
        print(d, k, {c:"%.4g x%d"%(sum(v)/len(v),len(v)) for c,v in cs.items()})
PY
find gpurun_out -name "*kernel_trace*" -size +8M -delete; find gpurun_out -name "*counter_collection.csv" -size +8M -delete
