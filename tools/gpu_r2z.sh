#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_mfma4 tools/probe_mfma4.hip 2>&1 | grep -v warning | head -5
timeout 60 /tmp/probe_mfma4 | tee gpurun_out/probe_mfma4.txt | head -70
