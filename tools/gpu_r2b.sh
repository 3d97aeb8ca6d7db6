#!/bin/bash
# Round 2, session B: the whole -m gpu suite (incl. the in-library multi-GPU tests and the full-size C2 oracle run), and the C2 / C5-family
# bench lines with the new roofline blocks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02b}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke_$TAG.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 --durations=15 > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "== pytest: $(tail -1 gpurun_out/pytest_gpu_$TAG.log)"
grep -E "FAILED|ERROR" gpurun_out/pytest_gpu_$TAG.log | head -20
grep -A 18 "slowest" gpurun_out/pytest_gpu_$TAG.log | head -20
timeout 600 python bench.py --config C2 > gpurun_out/bench_c2_$TAG.json 2> gpurun_out/bench_c2_$TAG.err; echo "bench C2 exit $?"; cut -c1-1500 gpurun_out/bench_c2_$TAG.json
timeout 600 python bench.py --config C5 --rows 1000000 --pmc off --no-jref > gpurun_out/bench_c5_$TAG.json 2> gpurun_out/bench_c5_$TAG.err; echo "bench C5 exit $?"; cut -c1-1500 gpurun_out/bench_c5_$TAG.json
