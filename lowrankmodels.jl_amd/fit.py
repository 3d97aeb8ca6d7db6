"""fit! / fit (reference: src/fit.jl:8-31, src/algorithms/proxgrad.jl:34-220) on the MI355X engine.

Single GPU: one ``glrm_hip_fit`` call -- the whole alternating loop runs in the C library.
Several GPUs: one process per GPU (``torch.distributed``, backend nccl = RCCL over xGMI); this
module runs the outer loop of Appendix A, calls the step-level C entry points on the rank's
row/column shard and all-gathers the factor the half-step just updated.
There is no CPU fallback: without libglrm_hip.so the call raises.
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import _capi
from .convergence import ConvergenceHistory, update_ch
from .params import AbstractParams, HipProxGradParams, ProxGradParams, SparseProxGradParams


LAST_EXCHANGE_PROBE = None  # the last set-up probe of a ShardedFit in this process (diagnostics; tests)


def _engine_opts(params):
    g = lambda k, d: getattr(params, k, d)
    return dict(device_id=g("device_id", -1), profile=1 if g("profile", False) else 0,
                waves_row=g("waves_row", 0), waves_col=g("waves_col", 0), tiled=g("tiled", 0), quad_gram=1 if g("quad_gram", False) else 0,
                sum_order=1 if g("mode", "fast") == "reference_order" else 0)


def _should_stop(i, prev, obj, scaled_abs_tol, rel_tol):
    """src/algorithms/proxgrad.jl:210-213 (a negative decrease also stops; obj == 0 divides like Julia)."""
    dec = prev - obj
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.float64(dec) / np.float64(obj)
    return i > 10 and (dec < scaled_abs_tol or rel < rel_tol)


def partition(ptr, parts):
    """Contiguous blocks of segments for ``parts`` shards, balanced by nnz;
    equal-count blocks are preferred when they are within 2 % of the nnz balance (enables the
    in-place all-gather)."""
    nseg, nnz = len(ptr) - 1, int(ptr[-1])
    eq = [nseg * i // parts for i in range(parts + 1)]
    if nseg % parts == 0 and nnz > 0:
        worst = max(int(ptr[eq[i + 1]] - ptr[eq[i]]) for i in range(parts))
        if worst <= 1.02 * nnz / parts + 1:
            return eq
    if nnz == 0:
        return eq
    b = [int(np.searchsorted(ptr, nnz * i / parts, side="left")) for i in range(parts + 1)]
    b[0], b[-1] = 0, nseg
    for i in range(1, parts + 1):
        b[i] = max(b[i], b[i - 1])
    return b


def _ensure_handle(glrm, api, params, allow_dense=True):
    """The model's engine handle (created on first use, kept on the model: Omega and A stay on the device across fit! calls,
    README.md:337-346 warm starts, cross-validation drivers).  A change of the regularizers only replaces the descriptors
    (scale_regularizer!, regularization_path).  Returns (handle, hard key, soft key)."""
    use_dense = allow_dense and api.dense_ok and glrm.dense_eligible() and getattr(params, "dense", True)
    hard, soft = glrm._descriptor_key()
    key = (id(api), _engine_opts(params)["device_id"], _engine_opts(params)["quad_gram"], _engine_opts(params)["sum_order"], hard)
    cache = glrm._handle_cache
    if cache is not None and (cache[2] != key or cache[4] != use_dense):
        glrm.close()
        cache = None
    if cache is None:
        # a model built from a sparse matrix's pattern goes over as its column view alone (GLRM_PROBLEM_ROWS_FROM_COLS)
        h = api.create(glrm.problem_arrays(dense=use_dense, cols_only=getattr(glrm, "_pattern_from_csc", False) and not use_dense), **_engine_opts(params))
        glrm._handle_cache = (api, h, key, soft, use_dense)
    elif cache[3] != soft:
        # only the regularizers changed: keep Omega / A on the device
        from .regularizers import pack_regs
        api.set_regularizers(cache[1], pack_regs(glrm.rx), pack_regs(glrm.ry))
        glrm._handle_cache = cache[:3] + (soft,) + tuple(cache[4:])
    return glrm._handle_cache[1], key, soft


def fit_b(glrm, params=None, *, ch=None, verbose=True, engine=None, group=None, **kwargs):
    """``fit!(glrm, params; ch, verbose)`` -> ``(glrm.X, glrm.Y, ch)``; glrm.X / glrm.Y are updated in
    place (warm start: a second call continues, README.md:337-346).  ``params`` may also be passed
    as the keyword the reference's dispatcher looks for (src/fit.jl:8-12).

    ``engine`` is a test hook (an ``_capi.Api``); the product default is the HIP library.
    """
    if params is None:  # src/fit.jl:13-18: a SparseMatrixCSC model defaults to the sparse solver, everything else to prox-grad
        from .glrm import _issparse
        params = SparseProxGradParams() if _issparse(glrm.A) else HipProxGradParams()
    if not isinstance(params, AbstractParams):
        raise TypeError("params must be an AbstractParams (ProxGradParams / HipProxGradParams)")
    sparse = isinstance(params, SparseProxGradParams)
    if ch is None:
        ch = ConvergenceHistory("SparseProxGradGLRM" if sparse else "ProxGradGLRM")
    api = engine if engine is not None else _capi.hip_api()
    world = 1
    if group is not None or os.environ.get("WORLD_SIZE", "1") != "1":
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
    if world > 1:
        if sparse:
            raise NotImplementedError("SparseProxGradParams runs on a single shard (step-level gradstep_x / gradstep_y exist for hosts)")
        return _fit_distributed(glrm, params, ch, verbose, api, group)

    if np.linalg.norm(glrm.Y) == 0:
        raise ValueError("Y is all zeros (the reference cannot start from Y == 0, src/algorithms/proxgrad.jl:45-48)")
    if getattr(params, "ngpus", 1) > 1 and not sparse:
        return _fit_multi_in_process(glrm, params, ch, verbose, api)
    h = _ensure_handle(glrm, api, params)[0]
    X = np.asfortranarray(glrm.X, dtype=np.float64)
    Y = np.asfortranarray(glrm.Y, dtype=np.float64)
    if sparse:  # src/algorithms/sparse_proxgrad.jl:22-134
        if verbose:
            print(params)
            print("Fitting GLRM")
        obj, sec = api.fit_sparse(h, params, X, Y)
        glrm.X[...] = X
        glrm.Y[...] = Y
        for i in range(len(obj)):
            update_ch(ch, sec[i] - (sec[i - 1] if i else 0.0), float(obj[i]))
        return glrm.X, glrm.Y, ch
    if verbose:
        print("Fitting GLRM")
    obj, sec = api.fit(h, params, X, Y)
    if X is not glrm.X:
        glrm.X[...] = X
    if Y is not glrm.Y:
        glrm.Y[...] = Y
    scaled_abs_tol = params.abs_tol * float(glrm._rowptr[-1])
    for i in range(len(obj)):
        update_ch(ch, sec[i] - (sec[i - 1] if i else 0.0), float(obj[i]))
        if verbose and i >= 1 and i % 10 == 0 and not _should_stop(i, obj[i - 1], obj[i], scaled_abs_tol, params.rel_tol):
            print(f"Iteration {i}: objective value = {obj[i]}")
    return glrm.X, glrm.Y, ch


def _record(glrm, params, ch, verbose, obj, sec):
    scaled_abs_tol = params.abs_tol * float(glrm._rowptr[-1])
    for i in range(len(obj)):
        update_ch(ch, sec[i] - (sec[i - 1] if i else 0.0), float(obj[i]))
        if verbose and i >= 1 and i % 10 == 0 and not _should_stop(i, obj[i - 1], obj[i], scaled_abs_tol, params.rel_tol):
            print(f"Iteration {i}: objective value = {obj[i]}")


def _fit_multi_in_process(glrm, params, ch, verbose, api):
    """``HipProxGradParams(ngpus=N)``: one process, N devices (include/glrm_hip.h, glrm_hip_multi_*).  The multi handle is cached on
    the model like the single-device one (Omega / A stay on the devices across warm starts; new regularizers only replace the
    descriptors)."""
    from .regularizers import pack_regs
    use_dense = api.dense_ok and glrm.dense_eligible() and getattr(params, "dense", True)
    hard, soft = glrm._descriptor_key()
    key = (id(api), "multi", params.ngpus, tuple(params.device_ids or ()), params.exchange, params.x_chunks, bool(getattr(params, "quad_gram", False)), getattr(params, "mode", "fast"), hard)
    cache = glrm._handle_cache
    if cache is not None and (cache[2] != key or cache[4] != use_dense):
        glrm.close()
        cache = None
    if cache is None:
        o = _engine_opts(params)
        mh = api.multi_create(glrm.problem_arrays(dense=use_dense), params.ngpus, params.device_ids, 1 if params.exchange == "rccl" else 0,
                              params.x_chunks, profile=o["profile"], waves_row=o["waves_row"], waves_col=o["waves_col"], tiled=o["tiled"], quad_gram=o["quad_gram"],
                              sum_order=o["sum_order"])
        glrm._handle_cache = (api, mh, key, soft, use_dense, "multi")
    elif cache[3] != soft:
        api.multi_set_regularizers(cache[1], pack_regs(glrm.rx), pack_regs(glrm.ry))
        glrm._handle_cache = cache[:3] + (soft,) + tuple(cache[4:])
    mh = glrm._handle_cache[1]
    X = np.asfortranarray(glrm.X, dtype=np.float64)
    Y = np.asfortranarray(glrm.Y, dtype=np.float64)
    if verbose:
        print("Fitting GLRM")
    obj, sec = api.multi_fit(mh, params, X, Y)
    if X is not glrm.X:
        glrm.X[...] = X
    if Y is not glrm.Y:
        glrm.Y[...] = Y
    _record(glrm, params, ch, verbose, obj, sec)
    return glrm.X, glrm.Y, ch


def fit(glrm, *args, **kwargs):
    """Non-mutating ``fit`` (src/fit.jl:24-31): returns (X', Y, ch) and restores glrm.X / glrm.Y."""
    X0, Y0 = glrm.X.copy(order="F"), glrm.Y.copy(order="F")
    X, Y, ch = fit_b(glrm, *args, **kwargs)
    Xo, Yo = X.copy(order="F"), Y.copy(order="F")
    glrm.X[...] = X0
    glrm.Y[...] = Y0
    return Xo.T, Yo, ch


def objective(glrm, X=None, Y=None, *, include_regularization=True, engine=None, **_):
    """objective(glrm, X, Y; include_regularization) over observed_examples (src/evaluate_fit.jl:57-83)."""
    api = engine if engine is not None else _capi.hip_api()
    X = np.asfortranarray(glrm.X if X is None else X, dtype=np.float64)
    Y = np.asfortranarray(glrm.Y if Y is None else Y, dtype=np.float64)
    h = _ensure_handle(glrm, api, HipProxGradParams())[0]  # resident handle: no re-upload per evaluation
    return api.objective(h, X, Y, include_regularization)


# ----------------------------------------------------------------------------- multi-GPU host

class ShardedFit:
    """The outer loop of src/algorithms/proxgrad.jl:107-217 for one rank of a row/column-sharded fit.

    The rank owns rows [rb,re) (X half-step) and columns [cb,ce) (Y half-step); X and Y are
    replicated.  After the X half-step every rank all-gathers the X blocks, after the Y half-step
    the Y blocks and the per-column objectives; the recorded objective is a fixed-order sum of the
    gathered per-column values, so it does not depend on the number of ranks.
    """

    def __init__(self, api, prob, row_bounds, col_bounds, group=None, device=None, stream=None, opts=None, x_chunks=1, whole_signature=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.api, self.group = torch, dist, api, group
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:  # single process: the gathers below are no-ops
            self.rank, self.world = 0, 1
        self.row_bounds, self.col_bounds = list(row_bounds), list(col_bounds)
        self.m, self.n, self.k = prob.m, prob.n, prob.k
        ys = prob.ystart
        self.d = int(ys[-1])                                       # vectors of Y (> n with multi-dimensional losses)
        self.y_bounds = [int(ys[c]) for c in self.col_bounds]      # the rank's block of Y in vector units
        self.device = device if device is not None else torch.device("cpu")
        o = dict(opts or {})
        if self.device.type == "cuda":
            o["device_id"] = self.device.index if self.device.index is not None else torch.cuda.current_device()
        # One shard of a sharded fit: upload only, then choose the kernel families from the signature of the WHOLE problem (the
        # families add in different orders; every rank must land on the same one -- include/glrm_hip.h, glrm_signature).  The
        # all-gather below is set-up traffic (six integers per rank), not a data-path collective.
        # `whole_signature`: a caller that already knows the whole problem's signature (bench.py --emulate-rank: one process holding
        # ONE shard of an N-shard problem) passes it instead.
        sharded = self.world > 1 or whole_signature is not None
        self.h = api.create(prob, stream=stream, defer=sharded, **o)
        if whole_signature is not None:
            api.finalize(self.h, whole_signature)
        elif self.world > 1:
            mine = api.signature(self.h).astuple()
            parts = [None] * self.world
            dist.all_gather_object(parts, mine, group=group)
            api.finalize(self.h, _capi.CSignature.combine(parts))
        self.ld = api.factor_ld(self.h)
        z = lambda cnt: torch.zeros(cnt, dtype=torch.float64, device=self.device)
        self.dX, self.dY, self.dObjCol, self.dObjRow = z(self.m * self.ld), z(self.d * self.ld), z(self.n), z(self.m)
        api.bind_buffers(self.h, self.dX.data_ptr(), self.dY.data_ptr(), self.dObjCol.data_ptr(), self.dObjRow.data_ptr())
        # Pipelined X exchange: the X half-step runs in `x_chunks` row chunks; the all-gather of a finished chunk
        # proceeds on a side stream while the next chunk is swept (rows are independent).  Needs equal row blocks.
        nrows = [row_bounds[r + 1] - row_bounds[r] for r in range(self.world)]
        self.x_chunks = int(x_chunks) if (self.world > 1 and x_chunks > 1 and len(set(nrows)) == 1 and nrows[0] % x_chunks == 0) else 1
        # Arrival order (include/glrm_hip.h: glrm_hip_step_y_arrival): the Y half-step is told which rows of X each chunk exchange
        # fills and the event behind it, instead of waiting for the whole pipelined exchange between the half-steps; GLRM_ARRIVAL=0
        # restores the wait
        self._arrival = self.x_chunks > 1 and os.environ.get("GLRM_ARRIVAL", "1") != "0"
        self._arrival_blocks, self._arrival_events = None, []
        if self.x_chunks > 1:
            c = nrows[0] // self.x_chunks
            self._stage = [torch.empty(self.world * c * self.ld, dtype=torch.float64, device=self.device) for _ in range(self.x_chunks)]
            self._comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        # GLRM_GATHER: how the updated blocks reach the other ranks.
        #   allgather  one in-place all_gather_into_tensor (RCCL picks the algorithm: on a ring the 1/N slice travels N - 1 hops)
        #   p2p        every owner sends its block straight to every peer in ONE group of point-to-point operations (ncclSend / ncclRecv
        #              on RCCL): all seven xGMI links of a GPU carry a block at once -- the direct exchange of DESIGN.md section 6
        #   broadcast  one broadcast per owner (what ragged blocks fall back to)
        #   auto       nccl with more than two ranks: BOTH are timed once at set-up on the X block (a few ms each) and the faster one is
        #              kept on every rank (the probe is published as `exchange_probe_ms`); otherwise all-gather on nccl, broadcasts elsewhere
        mode = os.environ.get("GLRM_GATHER", "auto")
        self._p2p = self.world > 1 and mode == "p2p"
        # ragged blocks (nnz-balanced partitions of real data): one grouped point-to-point exchange instead of one broadcast per owner
        # whenever the backend is RCCL and nothing else was asked for (every transfer carries its own size)
        self._ragged_p2p = self.world > 1 and mode == "auto" and self.device.type == "cuda" and dist.get_backend(group) == "nccl"
        self._inplace_ok = self.world > 1 and mode not in ("broadcast", "p2p") and (mode == "allgather" or dist.get_backend(group) == "nccl")
        self.exchange_ms = {"x": 0.0, "y": 0.0, "objective": 0.0}   # time the rank's stream spent in the exchanges (profile runs)
        self._timed = []                                            # (kind, start event, end event) awaiting a synchronisation
        self._profile = bool(o.get("profile")) and self.device.type == "cuda"
        self.exchange_probe_ms = None
        # (GLRM_GATHER_PROBE=1 runs the probe on any backend: how the gloo tests cover this code before the first multi-GPU node does)
        on_rccl = self.device.type == "cuda" and self.world > 1 and dist.get_backend(group) == "nccl"
        if self.world > 2 and mode == "auto" and (on_rccl or os.environ.get("GLRM_GATHER_PROBE") == "1"):
            self._probe_exchange()

    def _probe_exchange(self):
        """Time one all-gather and one point-to-point exchange of the X blocks (contents: whatever dX holds, it is rewritten by
        set_factors afterwards) and keep the faster on every rank; a failing p2p path leaves the all-gather in place."""
        torch, dist = self.torch, self.dist
        res = {}
        sync = (lambda: torch.cuda.synchronize(self.device)) if self.device.type == "cuda" else (lambda: None)
        for name in ("allgather", "p2p"):
            self._p2p = name == "p2p"
            # Every rank runs the SAME sequence of collectives whatever fails where: only the exchange under test sits inside the try, and
            # (ok, time) is reduced unconditionally afterwards -- a rank whose p2p leg throws still pairs its all_reduce with the others'.
            ok, dt, err = 1.0, float("inf"), None
            for rep in range(2):  # first repetition: connection set-up
                sync()
                dist.barrier(group=self.group)
                t0 = time.perf_counter()
                try:
                    self._gather(self.dX, self.row_bounds, self.ld)
                    sync()
                except Exception as e:  # noqa: BLE001 -- an unsupported path must not take the fit down
                    ok, err = 0.0, repr(e)
                dt = (time.perf_counter() - t0) * 1e3
            t = torch.tensor([dt if ok else 0.0, -ok], dtype=torch.float64, device=self.device)   # MAX of (time, -ok): any failure -> -ok = 0
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            res[name] = float(t[0].item()) if float(t[1].item()) < -0.5 else float("inf")
            if err is not None:
                res[name + "_error"] = err
        flag = torch.tensor([1.0 if res["p2p"] < res["allgather"] else 0.0], dtype=torch.float64, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)   # every rank must have seen p2p win
        self._p2p = bool(flag.item() > 0.5)
        res["chosen"] = "p2p" if self._p2p else "allgather"
        self.exchange_probe_ms = res
        global LAST_EXCHANGE_PROBE
        LAST_EXCHANGE_PROBE = res

    def close(self):
        if self.h is not None:
            self.api.destroy(self.h)
            self.h = None

    def _timed_gather(self, kind, buf, bounds, unit):
        """_gather bracketed by events on the rank's stream when the handle profiles (bench runs): the collectives are enqueued on the
        stream the sweeps run on, so the elapsed time between the events is what the exchange added to the iteration."""
        if not (self._profile and self.world > 1):
            return self._gather(buf, bounds, unit)
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record()
        self._gather(buf, bounds, unit)
        b.record()
        self._timed.append((kind, a, b))

    def exchange_times(self):
        """Summed exchange time per kind in ms since the last call (synchronises)."""
        if self._timed:
            self.torch.cuda.synchronize(self.device)
            for kind, a, b in self._timed:
                self.exchange_ms[kind] += a.elapsed_time(b)
            self._timed = []
        out, self.exchange_ms = self.exchange_ms, {"x": 0.0, "y": 0.0, "objective": 0.0}
        return out

    def _gather(self, buf, bounds, unit):
        """Make ``buf`` (global length) identical on every rank: rank r contributed
        buf[bounds[r]*unit : bounds[r+1]*unit].  Equal blocks -> one in-place all-gather (each GPU
        pushes its 1/G slice to its peers); ragged blocks -> one grouped point-to-point exchange on RCCL, one broadcast per owner elsewhere."""
        if self.world == 1:
            return
        dist, sizes = self.dist, [(bounds[r + 1] - bounds[r]) * unit for r in range(self.world)]
        if self._p2p:  # ragged blocks are fine: every transfer carries its own size
            self._p2p_ranges(buf, [(bounds[r] * unit, bounds[r + 1] * unit) for r in range(self.world)])
            return
        if self._inplace_ok and len(set(sizes)) == 1 and sizes[0] > 0:
            own = buf[bounds[self.rank] * unit: bounds[self.rank + 1] * unit]
            # NCCL / RCCL allow sendbuff == recvbuff + rank * count; GLRM_GATHER_INPLACE=0 sends a copy of the block instead
            src = own if os.environ.get("GLRM_GATHER_INPLACE", "1") != "0" else own.clone()
            dist.all_gather_into_tensor(buf, src, group=self.group)
            return
        if self._ragged_p2p:
            self._p2p_ranges(buf, [(bounds[r] * unit, bounds[r + 1] * unit) for r in range(self.world)])
            return
        for r in range(self.world):
            if sizes[r]:
                src = dist.get_global_rank(self.group, r) if self.group is not None else r
                dist.broadcast(buf[bounds[r] * unit: bounds[r + 1] * unit], src=src, group=self.group)

    def _p2p_ranges(self, buf, ranges):
        """Rank r owns buf[ranges[r][0]:ranges[r][1]]: send the own range to every peer and receive every peer's range in place, as ONE
        group of point-to-point operations (grouped ncclSend / ncclRecv on RCCL)."""
        dist = self.dist
        g = lambda r: dist.get_global_rank(self.group, r) if self.group is not None else r
        lo, hi = ranges[self.rank]
        own = buf[lo:hi]
        ops = []
        for r in range(self.world):
            if r == self.rank:
                continue
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, own, g(r), group=self.group))
            if ranges[r][1] > ranges[r][0]:
                ops.append(dist.P2POp(dist.irecv, buf[ranges[r][0]:ranges[r][1]], g(r), group=self.group))
        if not ops:
            return
        # only ProcessGroupNCCL orders point-to-point transfers with the CUDA stream; gloo (the plumbing-check backend) reads and
        # writes device tensors from the host, so the stream is drained on both sides of the exchange there
        host_sync = self.device.type == "cuda" and dist.get_backend(self.group) != "nccl"
        if host_sync:
            self.torch.cuda.synchronize(self.device)
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if host_sync:
            self.torch.cuda.synchronize(self.device)

    def _step_x_pipelined(self, params):
        """Last inner X sweep in row chunks; the exchange of chunk j proceeds on a side stream while chunk j+1 is being swept.  With
        the point-to-point exchange every chunk goes straight from its owner into its place on every peer; with the all-gather it
        lands in a staging buffer (the ranks' chunks are not adjacent in X) and is copied into place."""
        torch, dist, api, h = self.torch, self.dist, self.api, self.h
        S, ld, rb = self.x_chunks, self.ld, self.row_bounds
        c = (rb[self.rank + 1] - rb[self.rank]) // S
        cuda = self.device.type == "cuda"
        blocks, events = [(rb[self.rank], rb[self.rank + 1], None)], []
        for j in range(S):
            api.step_x_range(h, j * c, (j + 1) * c, params.min_stepsize)
            own = self.dX[(rb[self.rank] + j * c) * ld: (rb[self.rank] + (j + 1) * c) * ld]
            if cuda:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(ev)
                    if self._p2p:
                        self._p2p_ranges(self.dX, [((rb[q] + j * c) * ld, (rb[q] + (j + 1) * c) * ld) for q in range(self.world)])
                    else:
                        dist.all_gather_into_tensor(self._stage[j], own, group=self.group)
                        for q in range(self.world):
                            if q != self.rank:
                                self.dX[(rb[q] + j * c) * ld: (rb[q] + (j + 1) * c) * ld].copy_(self._stage[j][q * c * ld: (q + 1) * c * ld], non_blocking=True)
                    if self._arrival:  # chunk j of every peer is in place once this event has fired
                        done = torch.cuda.Event()
                        done.record(self._comm_stream)
                        events.append(done)
            elif self._p2p:
                self._p2p_ranges(self.dX, [((rb[q] + j * c) * ld, (rb[q] + (j + 1) * c) * ld) for q in range(self.world)])
            else:
                dist.all_gather_into_tensor(self._stage[j], own.clone(), group=self.group)
                for q in range(self.world):
                    if q != self.rank:
                        self.dX[(rb[q] + j * c) * ld: (rb[q] + (j + 1) * c) * ld].copy_(self._stage[j][q * c * ld: (q + 1) * c * ld])
            if self._arrival:
                ev = events[-1].cuda_event if cuda else None
                # peers in ring order from this rank (the order a direct exchange delivers in; with one event per chunk it only
                # matters as the order in which the super-tiles are launched)
                blocks += [(rb[q] + j * c, rb[q] + (j + 1) * c, ev) for q in ((self.rank + d) % self.world for d in range(1, self.world))]
        if self._arrival:  # the first inner Y sweep waits block by block (iteration: step_y_arrival); nothing to wait for here
            self._arrival_blocks, self._arrival_events = blocks, events
            return
        if cuda:
            if self._profile:  # what the pipeline did NOT hide: the wait for the last chunks
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                torch.cuda.current_stream().wait_stream(self._comm_stream)
                b.record()
                self._timed.append(("x", a, b))
            else:
                torch.cuda.current_stream().wait_stream(self._comm_stream)

    def initial_objective(self):
        api, h = self.api, self.h
        api.col_losses(h)
        self._gather(self.dObjCol, self.col_bounds, 1)
        loss = api.sum(h, self.dObjCol.data_ptr(), self.n)
        api.row_penalties(h)
        self._gather(self.dObjRow, self.row_bounds, 1)
        px = api.sum(h, self.dObjRow.data_ptr(), self.m)
        api.col_penalties(h)
        self._gather(self.dObjCol, self.col_bounds, 1)
        py = api.sum(h, self.dObjCol.data_ptr(), self.n)
        return loss + (px + py)  # err += calc_penalty(...), src/evaluate_fit.jl:19-21

    def iteration(self, params):
        """One outer iteration (src/algorithms/proxgrad.jl:107-205); returns sum(obj_by_col)."""
        api, h = self.api, self.h
        if params.inner_iter_X > 1 or params.inner_iter_Y > 1:
            api.reset_stepsizes(h, params.stepsize)
        for _ in range(params.inner_iter_X - 1):
            api.step_x(h, params.min_stepsize)
        if self.x_chunks > 1:
            self._step_x_pipelined(params)
        else:
            api.step_x(h, params.min_stepsize)
            self._timed_gather("x", self.dX, self.row_bounds, self.ld)  # inner X sweeps only touch own rows: gather once
        for it in range(params.inner_iter_Y):
            if it == 0 and self._arrival_blocks is not None:
                api.step_y_arrival(h, params.min_stepsize, self._arrival_blocks)  # returns with every event waited for on the stream
                self._arrival_blocks = None
            else:
                api.step_y(h, params.min_stepsize)
        self._timed_gather("y", self.dY, self.y_bounds, self.ld)
        self._timed_gather("objective", self.dObjCol, self.col_bounds, 1)
        return api.sum(h, self.dObjCol.data_ptr(), self.n)


def _fit_distributed(glrm, params, ch, verbose, api, group):
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    rbs, cbs = partition(glrm._rowptr, world), partition(glrm._colptr, world)
    use_dense = api.dense_ok and glrm.dense_eligible() and getattr(params, "dense", True)
    prob = glrm.problem_arrays(rows=(rbs[rank], rbs[rank + 1]), cols=(cbs[rank], cbs[rank + 1]), dense=use_dense)
    if api.device_type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
        stream = torch.cuda.current_stream().cuda_stream
    else:
        device, stream = torch.device("cpu"), None
    opts = _engine_opts(params)
    sf = ShardedFit(api, prob, rbs, cbs, group=group, device=device, stream=stream, opts=opts,
                    x_chunks=int(os.environ.get("GLRM_X_CHUNKS", getattr(params, "x_chunks", 1))))
    try:
        if np.linalg.norm(glrm.Y) == 0:
            raise ValueError("Y is all zeros (the reference cannot start from Y == 0)")
        X = np.asfortranarray(glrm.X, dtype=np.float64)
        Y = np.asfortranarray(glrm.Y, dtype=np.float64)
        api.set_factors(sf.h, X, Y)
        api.reset_stepsizes(sf.h, params.stepsize)
        scaled_abs_tol = params.abs_tol * float(glrm._rowptr[-1])
        if verbose and rank == 0:
            print("Fitting GLRM")
        update_ch(ch, 0, sf.initial_objective())
        t = time.time()
        for i in range(1, params.max_iter + 1):
            obj = sf.iteration(params)
            t = time.time() - t
            update_ch(ch, t, obj)
            t = time.time()
            if _should_stop(i, ch.objective[-2], obj, scaled_abs_tol, params.rel_tol):
                break
            if verbose and rank == 0 and i % 10 == 0:
                print(f"Iteration {i}: objective value = {ch.objective[-1]}")
        api.get_factors(sf.h, X, Y)
        glrm.X[...] = X
        glrm.Y[...] = Y
    finally:
        sf.close()
    return glrm.X, glrm.Y, ch
