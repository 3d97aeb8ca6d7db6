"""Run bench.py against another build of the engine in the package directory (same-box A/B of kernel variants):
python tests/perf/ab_lib.py libglrm_hip_cur.so --config C5 --steps 5 ..."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lowrankmodels.jl_amd._capi as c
c.HIP_LIB_PATH = os.path.join(os.path.dirname(c.__file__), sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
