#!/bin/bash
# round 2, session 3h: HBM-side traffic of the cached gather row sweep at C4 (FETCH_SIZE / WRITE_SIZE, separate passes)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
out=gpurun_out/cachedpmc
rm -rf $out; mkdir -p $out
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$out/pmc_$C -o pmc -- python $R/bench.py --config C4 --steps 2 --warmup 2 --no-jref --no-cpu-baseline --no-convergence-run --pmc off > /dev/null 2> $R/$out/$C.err); echo "pmc $C exit $?"
done
python tools/pmc_summary.py $out > $out/summary.md
find $out -name "*.csv" -size +2M -delete
grep -i "regcached\|counter" $out/summary.md | head
