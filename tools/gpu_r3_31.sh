#!/bin/bash
# round 3, call 31: SQ counters of the heterogeneous LDS-tiled kernels on the C5-family shape after the DMA staging (two PMC passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
Q="--config C5 --rows 1000000 --pmc off --no-jref --no-cpu-baseline --no-convergence-run --steps 2 --warmup 1"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES"
i=0
for C in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_t$i
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_t$i -o pmc -- python $R/bench.py $Q > /dev/null 2> /tmp/pmc_t$i.err)
  python - "$i" <<'PY'
import csv, glob, collections, sys, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"/tmp/pmc_t{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        kn=r["Kernel_Name"]
        if "tiled_sweep_kernel" in kn or "tiled_col_pass_kernel" in kn:
            key = re.sub(r"^void glrm::", "", kn)[:64]
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    v={c: sum(x)/len(x) for c,x in rows[k].items()}
    n=len(next(iter(rows[k].values())))
    print("SQ", k, "dispatches", n, {c: "%.4g" % x for c,x in sorted(v.items())})
PY
done 2>&1 | tee gpurun_out/r3_31_sq.txt
