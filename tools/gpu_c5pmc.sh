#!/bin/bash
# SQ / LDS counters of the heterogeneous (C5-family) tiled sweeps: two rocprofv3 --pmc passes, summarised per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/c5pmc; mkdir -p gpurun_out/c5pmc
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/c5pmc/pmc_$tag -o pmc -- python $R/bench.py --config C5 --rows-per-gpu 1000000 --steps 2 --warmup 2 --no-cpu-baseline --no-convergence-run > /dev/null 2> $R/gpurun_out/c5pmc/$tag.err); echo "pmc $tag exit $?"
done
python tools/pmc_summary.py gpurun_out/c5pmc > gpurun_out/c5pmc/summary.md
find gpurun_out/c5pmc -name "*.csv" -size +2M -delete
cat gpurun_out/c5pmc/summary.md
