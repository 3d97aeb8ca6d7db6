#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mfma tools/ubench_mfma.hip 2>&1 | grep -v warning | head -5
timeout 120 /tmp/ubench_mfma | tee gpurun_out/ubench_mfma.txt
