#!/bin/bash
# round 3, call 22: the three soak failures (seeds 1070, 1148, 1189): this round's MultinomialOrdinal formulas or older? where do they leave the oracle?
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
V=$PWD/lowrankmodels.jl_amd/libglrm_hip_mnlordlibm.so
{
for S in 1070 1148 1189; do
  echo "=== seed $S product"; timeout 100 python tests/perf/soak_fuzz.py $S $((S+1)) | head -3
  echo "=== seed $S -DGLRM_MNLORD_LIBM"; GLRM_HIP_LIB_PATH=$V timeout 100 python tests/perf/soak_fuzz.py $S $((S+1)) | head -3
  echo "=== seed $S steps (product)"; timeout 100 python tests/perf/dbg_fuzz_steps.py $S 8 2>&1 | head -60
done
} > gpurun_out/r3_22_soak_dbg.txt 2>&1
head -150 gpurun_out/r3_22_soak_dbg.txt
