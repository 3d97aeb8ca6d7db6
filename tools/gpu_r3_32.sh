#!/bin/bash
# round 3, call 32: state after the DMA staging / ordinal kind class: smoke, the whole -m gpu suite, a short C4 line (kernels untouched: regression check)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r3_32
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/${T}_smoke.log
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pytest.log; grep -E "FAILED|ERROR" gpurun_out/${T}_pytest.log | head
timeout 600 python bench.py --steps 10 --warmup 3 --pmc off --no-jref --no-cpu-baseline --no-convergence-run > gpurun_out/${T}_bench_c4_short.json 2> gpurun_out/${T}_bench_c4_short.err; echo "bench C4 exit $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_32_bench_c4_short.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("C4 short: ms/step %.1f row %.2f col %.2f value %.4g frac %.3f" % (d["ms_per_step"],k["row_sweep_ms"],k["col_sweep_ms"],d["value"],d["roofline"]["frac"]))
PY
