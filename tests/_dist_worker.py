"""Worker for tests/test_distributed_gloo.py: one rank of a row/column-sharded fit over gloo (CPU).
The engine is the CPU oracle (test hook); the host loop, the sharding and the collectives are the
product code that runs unchanged over RCCL on GPUs."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import cases  # noqa: E402
import lowrankmodels.jl_amd as L  # noqa: E402
import oracle as O  # noqa: E402

if __name__ == "__main__":
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    res = {}
    for name in sys.argv[2:]:
        kwargs, params = cases.build_golden_case(name)
        g = L.GLRM(**kwargs)
        X, Y, ch = L.fit_b(g, params, verbose=False, engine=O.oracle_api())
        res[name + "_obj"] = np.array(ch.objective)
        res[name + "_X"] = X
        res[name + "_Y"] = Y
    _fit = sys.modules["lowrankmodels.jl_amd.fit"]  # (the package exports the function `fit` under the module's name)
    if _fit.LAST_EXCHANGE_PROBE is not None:  # GLRM_GATHER_PROBE=1: which exchange the set-up probe kept (must agree on every rank)
        res["probe_chose_p2p"] = np.array(_fit.LAST_EXCHANGE_PROBE["chosen"] == "p2p")
        res["probe_ms"] = np.array([_fit.LAST_EXCHANGE_PROBE["allgather"], _fit.LAST_EXCHANGE_PROBE["p2p"]])
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()
