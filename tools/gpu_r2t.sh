#!/bin/bash
# round 2, session t: LDS / SQ counters of the tiled sweeps at C2, conflict-free tile reads vs padded build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
for lib in libglrm_hip.so libglrm_hip_norot.so; do
  out=gpurun_out/rotpmc_${lib%.so}
  rm -rf $out; mkdir -p $out
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
    tag=$(echo $C | tr ' ' '_' | cut -c1-24)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$out/pmc_$tag -o pmc -- python $R/tests/perf/ab_lib.py $lib --config C2 --steps 2 --warmup 2 --no-jref --no-cpu-baseline --no-convergence-run --pmc off > /dev/null 2> $R/$out/$tag.err); echo "pmc $lib $tag exit $?"
  done
  python tools/pmc_summary.py $out > $out/summary.md
  find $out -name "*.csv" -size +2M -delete
  echo "=== $lib"; cat $out/summary.md
done
