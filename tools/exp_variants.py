"""Run bench.py under a grid of tuning knobs (env vars / flags) and tabulate the sweep-kernel times."""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env_extra, args):
    env = dict(os.environ, **{k: str(v) for k, v in env_extra.items()})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines()[::-1]:
        if line.startswith("{"):
            return json.loads(line)
    return {"error": r.stderr[-400:]}


if __name__ == "__main__":
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    extra = sys.argv[2:]
    grid = []
    for G in (4, 8, 16):
        for ur, uc in ((1, 1), (2, 2)):
            for wr, wc in ((1, 4), (1, 8), (4, 4)):
                grid.append((G, ur, uc, wr, wc))
    for G, ur, uc, wr, wc in grid:
        d = run({"GLRM_HIP_LANES_PER_OBS": G, "GLRM_HIP_UNROLL_ROW": ur, "GLRM_HIP_UNROLL_COL": uc},
                ["--waves-row", str(wr), "--waves-col", str(wc)] + extra)
        k = d.get("kernels", {})
        line = dict(G=G, unroll_row=ur, unroll_col=uc, waves_row=wr, waves_col=wc, row_ms=k.get("row_sweep_ms"),
                    col_ms=k.get("col_sweep_ms"), ms_per_step=d.get("ms_per_step"), err=d.get("error"))
        out.write(json.dumps(line) + "\n")
        out.flush()
