"""python tests/perf/rerun_seed.py SEED [REPEATS] [FIRST]: soak_fuzz.one(SEED) repeated (is a finding reproducible?), optionally after running the
seeds FIRST .. SEED-1 in the same process (does it depend on what ran before?)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tests", "perf", "soak_fuzz.py"))
soak = importlib.util.module_from_spec(spec)
spec.loader.exec_module(soak)
seed = int(sys.argv[1])
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
soak.O.set_threads(4)
if len(sys.argv) > 3:
    for s in range(int(sys.argv[3]), seed):
        try:
            soak.one(s)
        except Exception as e:  # noqa: BLE001
            print("seed", s, "raised", repr(e)[:100])
for _ in range(rep):
    print(seed, soak.one(seed), flush=True)
