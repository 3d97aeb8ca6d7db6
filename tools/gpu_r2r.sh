#!/bin/bash
# round 2, session r: row-per-lane tiled sweep prototype (tools/ubench_lanerow.hip) next to the product's tiled sweep at the same shape
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ubench_lanerow tools/ubench_lanerow.hip 2> gpurun_out/lanerow_build.log || { tail gpurun_out/lanerow_build.log; exit 1; }
timeout 300 /tmp/ubench_lanerow 19 | tee gpurun_out/ubench_lanerow.txt

