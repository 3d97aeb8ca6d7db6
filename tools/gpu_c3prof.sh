#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof_c3 gpurun_out/pmcd_*
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o bench -- python $R/bench.py --config C3 --steps 4 --warmup 2 > $R/gpurun_out/prof_c3_bench.json 2> $R/gpurun_out/prof_c3.err); echo "prof exit $?"
head -8 gpurun_out/prof_c3/bench_kernel_stats.csv
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcd_$tag -o pmc -- python $R/bench.py --config C3 --steps 2 --warmup 1 > $R/gpurun_out/pmcd_$tag.json 2> $R/gpurun_out/pmcd_$tag.err); echo "pmc $tag exit $?"
done
find gpurun_out -name "*kernel_trace*" -size +8M -delete
