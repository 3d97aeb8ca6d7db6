#!/bin/bash
# ONE parameterised GPU-box session runner (replaces the per-question gpu_r2*.sh / gpu_r3_*.sh scripts of rounds 2-3; what each of those
# sessions asked is listed in tools/SESSIONS.md).  Usage, from the build container:
#
#   gpurun --timeout 1500 -- 'tools/gpu_session.sh TAG STEP [STEP ...]'
#
# Every STEP is one quoted string "verb args..."; outputs land under gpurun_out/TAG/ (merged back by gpurun).  Verbs:
#   smoke                         __graft_entry__.smoke()
#   suite [pytest args]           python -m pytest tests -m gpu -q [args]                          -> pytest.log
#   tests NAME [pytest args]      the same for a selection, log under NAME                         -> pytest_NAME.log
#   bench NAME [bench.py args]    python bench.py args                                             -> bench_NAME.json / .err
#   kstats NAME [bench.py args]   rocprofv3 --kernel-trace --stats of bench.py args                -> kstats_NAME/ (csv), kstats_NAME.json
#   pmc NAME "CTR CTR" [args]     rocprofv3 --pmc CTR... of bench.py args (own run, no tracing)    -> pmc_NAME/ (csv)
#   py NAME script.py [args]      python script.py args                                            -> NAME.log
#   ab NAME LIBTAG [bench args]   bench.py against libglrm_hip_LIBTAG.so (tests/perf/ab_lib.py)    -> ab_NAME_LIBTAG.json
#   sh NAME 'shell command'       anything else                                                    -> NAME.log
# A step never aborts the session (set +e); every step is wrapped in its own `timeout` (STEP_TIMEOUT, default 900 s).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
TAG=${1:?tag}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
T=${STEP_TIMEOUT:-900}
for step in "$@"; do
  set -- $step
  verb=$1; shift
  t0=$(date +%s)
  case $verb in
    smoke)  timeout $T python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -1 $OUT/smoke.log ;;
    suite)  timeout $T python -m pytest tests -m gpu -q --durations=8 "$@" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $OUT/pytest.log
            tail -3 $OUT/pytest.log; grep -E "^(FAILED|ERROR)|fuzz accounting|\[jref" $OUT/pytest.log | cut -c1-600 | head -20 ;;
    tests)  n=$1; shift; timeout $T python -m pytest -m gpu -q --durations=5 "$@" > $OUT/pytest_$n.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $OUT/pytest_$n.log
            tail -3 $OUT/pytest_$n.log; grep -E "^(FAILED|ERROR)|fuzz accounting|\[jref" $OUT/pytest_$n.log | cut -c1-1500 | head -20 ;;
    bench)  n=$1; shift; timeout $T python bench.py "$@" > $OUT/bench_$n.json 2> $OUT/bench_$n.err; rc=$?; cut -c1-1200 $OUT/bench_$n.json; [ $rc -ne 0 ] && tail -5 $OUT/bench_$n.err ;;
    kstats) n=$1; shift; rm -rf $OUT/kstats_$n
            (cd /tmp && timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_$n -o k -- python $R/bench.py "$@" > $OUT/kstats_$n.json 2> $OUT/kstats_$n.err); rc=$?
            find $OUT/kstats_$n -name "*kernel_trace*" -size +6M -delete; f=$(find $OUT/kstats_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-220 ;;
    pmc)    n=$1; ctrs=$(echo $2 | tr ',' ' '); shift 2; rm -rf $OUT/pmc_$n
            (cd /tmp && timeout $T rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$n -o pmc -- python $R/bench.py "$@" > $OUT/pmc_$n.json 2> $OUT/pmc_$n.err); rc=$?
            python tools/pmc_summary.py $OUT/pmc_$n > $OUT/pmc_$n.md 2>/dev/null; head -12 $OUT/pmc_$n.md | cut -c1-260
            find $OUT/pmc_$n -name "*.csv" -size +6M -delete ;;
    py)     n=$1; shift; timeout $T python "$@" > $OUT/$n.log 2>&1; rc=$?; tail -8 $OUT/$n.log | cut -c1-400 ;;
    ab)     n=$1; lib=$2; shift 2; timeout $T python tests/perf/ab_lib.py libglrm_hip_$lib.so "$@" > $OUT/ab_${n}_$lib.json 2> $OUT/ab_${n}_$lib.err; rc=$?; cut -c1-600 $OUT/ab_${n}_$lib.json ;;
    sh)     n=$1; shift; timeout $T bash -c "$*" > $OUT/$n.log 2>&1; rc=$?; tail -8 $OUT/$n.log | cut -c1-400 ;;
    *)      echo "unknown verb $verb"; rc=2 ;;
  esac
  echo "== [$TAG] $verb ${1:-} exit $rc ($(( $(date +%s) - t0 )) s)"
done
