#!/bin/bash
# round 3, call 17: TCC hit rate of the C4 kernels with the final launch geometry (separate PMC pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
D=/tmp/pmc_tcc; rm -rf $D
(cd /tmp && timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D -o pmc -- python $R/bench.py --steps 2 --warmup 1 --pmc off --no-jref --no-cpu-baseline --no-convergence-run > /dev/null 2> $D.err)
python - <<'PY' | tee gpurun_out/r3_17_c4_tcc.txt
import csv,glob,re,collections
acc=collections.defaultdict(lambda:collections.defaultdict(float)); cnt=collections.Counter()
for p in glob.glob("/tmp/pmc_tcc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(p)):
        k=re.sub(r"\(.*","",r["Kernel_Name"])[:72]
        if "sweep" in k or "pass" in k or "persist" in k:
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k]+=1
for k,v in acc.items():
    h,m=v["TCC_HIT_sum"],v["TCC_MISS_sum"]
    print(f"{k}: dispatches {cnt[k]//2}, TCC hits {h:.4g}, misses {m:.4g}, hit rate {h/(h+m):.3f}")
PY
