"""Lane-by-lane host simulation of how the engine's sweep kernels add a segment's terms (test infrastructure).

Written from the kernel sources, one simulated lane per list element -- csrc/glrm_hip.hip (sweep_pass, block_combine), csrc/glrm_cached.hip
(reg_pass, row_combine), csrc/glrm_tiled.hpp (tiled_pass: batches of G entries, the tile bound `hi`, re-anchoring, col_reduce_kernel) and
csrc/glrm_device.hpp (group_sum, across_groups_sum, reg_eval) -- so that the C restatement in oracle/glrm_oracle.c (eng_pass: "position in the
window modulo batch", "group q takes q, q + T, ...") is checked against the literal control flow it abbreviates: tests/test_sum_order.py.
fma is exact rational arithmetic rounded once (Fraction -> float rounds to nearest even), i.e. IEEE fma for finite operands.
"""
from fractions import Fraction

INT_MAX = 0x7FFFFFFF


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def quad_loss(scale, u, a):  # src/losses.jl:144,146 as both sides write it
    d = u - a
    return scale * (d * d), 2 * d * scale


def lane_partials(x, y, k, G, R, rot=0):
    """p[j]: lane j's fma chain over its components, register i holding chunk i ^ rot (Vec<G, R>, tile_rot)."""
    p = []
    for j in range(G):
        s = 0.0
        for i in range(R // 2):
            c = (i ^ rot) * 2 * G + 2 * j
            for cc in (c, c + 1):
                s = fma(x[cc] if cc < k else 0.0, y[cc] if cc < k else 0.0, s)
        p.append(s)
    return p


def xor_step(v, d):
    return [v[j] + v[j ^ d] for j in range(len(v))]


def group_sum_mirrors(v):
    """glrm_device.hpp group_sum<G>: xor 1, xor 2, half mirror (i <-> 7 - i), mirror (i <-> 15 - i); returns lane 0's value"""
    G = len(v)
    if G >= 2:
        v = xor_step(v, 1)
    if G >= 4:
        v = xor_step(v, 2)
    if G >= 8:
        v = [v[j] + v[(j & ~7) + 7 - (j & 7)] for j in range(G)]
    if G >= 16:
        v = [v[j] + v[15 - j] for j in range(G)]
    assert all(a == v[0] for a in v)  # every lane of the group ends with the same bits
    return v[0]


def across_groups(vals64, G):
    """across_groups_sum<G>: v += shfl_xor(v, d) for d = G, 2G, ... 32 over a 64-lane wave whose groups hold replicated values"""
    v = list(vals64)
    d = G
    while d < 64:
        v = [v[l] + v[l ^ d] for l in range(64)]
        d <<= 1
    assert all(a == v[0] for a in v)
    return v[0]


def strided_pass(idx, vals, xv, fac, k, G, R, waves, scale, grad, scatter=False, U=None):
    """sweep_pass + block_combine (reg_pass + row_combine): lane group gg of wave w handles observations t == gg (mod TG), ascending."""
    NG, kp = 64 // G, G * R
    TG = NG * waves
    n = len(idx)
    waveJ, waveg = [], []
    for w in range(waves):
        J64 = [0.0] * 64
        g64 = [[0.0] * kp for _ in range(64)]
        for gi in range(NG):
            gg = w * NG + gi
            mine = list(range(gg, n, TG))
            if scatter:  # four observations per trip, lane u evaluates observation u; J is a lane partial summed over the group after the loop
                Jl = [0.0] * G
                gacc = [0.0] * kp
                for t0 in range(0, len(mine), 4):
                    trip = mine[t0:t0 + 4]
                    dl = []
                    for u, t in enumerate(trip):
                        p = lane_partials(xv, fac[idx[t]], k, G, R)
                        q = xor_step(xor_step(p, 1), 2)  # reduce-scatter pairings of group_sum: the same bits
                        L, dL = quad_loss(scale, q[0], vals[t])
                        Jl[u] += L
                        dl.append((dL, t))
                    for dL, t in dl:
                        y = fac[idx[t]]
                        for c in range(k):
                            gacc[c] = fma(dL, y[c], gacc[c])
                Jg = group_sum_mirrors(Jl)
            else:
                Jg, gacc = 0.0, [0.0] * kp
                for t in mine:
                    y = fac[idx[t]]
                    u_ = group_sum_mirrors(lane_partials(xv, y, k, G, R))
                    L, dL = quad_loss(scale, u_, vals[t])
                    Jg += L
                    if grad:
                        for c in range(k):
                            gacc[c] = fma(dL, y[c], gacc[c])
            for j in range(G):
                J64[gi * G + j] = Jg
                g64[gi * G + j] = gacc
        waveJ.append(across_groups(J64, G))
        waveg.append([across_groups([g64[l][c] for l in range(64)], G) for c in range(k)] if grad else None)
    if waves == 1:
        return waveJ[0], waveg[0]
    J, g = 0.0, [0.0] * k
    for w in range(waves):  # block_combine: in wave order from 0.0
        J += waveJ[w]
        if grad:
            g = [g[c] + waveg[w][c] for c in range(k)]
    return J, (g if grad else None)


def tiled_pass(idx, vals, xv, fac, k, G, R, TILE, tile_begin, tile_end, n_other, scale, grad, four=False, L2=False, rot=0, lossfn=None):
    """tiled_pass<G, R, ..., LOSS, GRAD, L2> of ONE lane group over tiles [tile_begin, tile_end): returns (J, g) of those tiles."""
    end = len(idx)
    lossfn = lossfn or (lambda c, u, a: quad_loss(scale, u, a))
    pos = 0
    while pos < end and idx[pos] < tile_begin * TILE:  # lower_bound_idx (rows: tile_begin = 0)
        pos += 1

    def batch_at(p):
        return [idx[p + j] if p + j < end else INT_MAX for j in range(G)]

    cb = batch_at(pos)
    J = [0.0] * G
    kp = G * R
    g = [0.0] * kp
    for t in range(tile_begin, tile_begin + 1 if L2 else tile_end):
        lo = t * TILE
        hi = min(tile_end * TILE if L2 else lo + TILE, n_other)
        done = False
        while not done:
            nxt_pos = pos + G
            nproc = 0
            if four:
                ok = []
                for u in range(G):
                    ok.append((u == 0 or ok[u - 1]) and cb[u] < hi)
                if ok[0]:
                    dL = [0.0] * G
                    for u in range(G):
                        if ok[u]:
                            y = fac[cb[u]]
                            p = lane_partials(xv, y, k, G, R, rot)
                            d = 1
                            while d < G:  # reduce-scatter: the pairings of the plain butterfly
                                p = xor_step(p, d)
                                d <<= 1
                            L, dL[u] = lossfn(cb[u], p[0], vals[pos + u])
                            J[u] += L
                    if grad:
                        for u in range(G):  # list order; observations past the window carry a zero derivative
                            y = fac[cb[u] if ok[u] else cb[0]]
                            for c in range(k):
                                g[c] = fma(dL[u], y[c], g[c])
                    nproc = sum(ok)
                    if not ok[G - 1]:
                        done = True
                else:
                    done = True
            else:
                for u0 in range(0, G, 2):
                    c0, c1 = cb[u0], cb[u0 + 1]
                    ok0 = (not done) and c0 < hi
                    ok1 = ok0 and c1 < hi
                    if ok0:
                        y0, y1 = fac[c0], fac[c1 if ok1 else c0]
                        p0, p1 = lane_partials(xv, y0, k, G, R, rot), lane_partials(xv, y1, k, G, R, rot)
                        dot = [(p1[j] if j & 1 else p0[j]) + (p1[j ^ 1] if j & 1 else p0[j ^ 1]) for j in range(G)]  # keep + dpp_xor1(send)
                        d = 2
                        while d < G:
                            dot = xor_step(dot, d)
                            d <<= 1
                        L0, d0 = lossfn(c0, dot[0], vals[pos + nproc])
                        L1, d1 = lossfn(c1 if ok1 else c0, dot[1 if G > 1 else 0], vals[pos + nproc + 1] if ok1 else vals[pos + nproc])
                        if not ok1:
                            L1, d1 = 0.0, 0.0
                        for j in range(G):
                            J[j] += L1 if j & 1 else L0
                        if grad:
                            for c in range(k):
                                g[c] = fma(d0, y0[c], g[c])
                            for c in range(k):
                                g[c] = fma(d1, y1[c], g[c])
                        nproc += 2 if ok1 else 1
                        if not ok1:
                            done = True
                    else:
                        done = True
            pos += nproc
            if not done:
                cb = batch_at(nxt_pos)
            elif nproc > 0:
                cb = batch_at(pos)
    v = J
    d = 1
    while d < G:  # group_sum (the xor steps; the mirrors of group_sum<G> pair the same partial sums)
        v = xor_step(v, d)
        d <<= 1
    Jout = v[0] * (1.0 if four else 2.0 / G)
    return Jout, g[:k]


def windowed_pass(idx, vals, xv, fac, k, G, R, TILE, tiles_per_sup, n_other, scale, grad, four=False, L2=False, rot=0, lossfn=None):
    """Row sweep (tiles_per_sup = 0: one pass over all tiles) or the column passes: one tiled_pass per super-tile, partial sums added in
    super-tile order from 0 (col_reduce_kernel / col_decide_kernel)."""
    ntiles = (n_other + TILE - 1) // TILE
    if tiles_per_sup <= 0:
        return tiled_pass(idx, vals, xv, fac, k, G, R, TILE, 0, ntiles, n_other, scale, grad, four, L2, rot, lossfn)
    nsup = (ntiles + tiles_per_sup - 1) // tiles_per_sup
    J, g = 0.0, [0.0] * k
    for s in range(nsup):
        tb = s * tiles_per_sup
        te = min(tb + tiles_per_sup, ntiles)
        Js, gs = tiled_pass(idx, vals, xv, fac, k, G, R, TILE, tb, te, n_other, scale, grad, four, L2, rot, lossfn)
        J += Js
        if grad:
            g = [g[c] + gs[c] for c in range(k)]
    return J, g


def reg_quad(scale, x, k, G, R):
    """reg_eval<G, R> for QuadReg: lane chains of fma(x, x, s), butterfly, times scale"""
    p = []
    for j in range(G):
        s = 0.0
        for i in range(R // 2):
            c = i * 2 * G + 2 * j
            for cc in (c, c + 1):
                a = x[cc] if cc < k else 0.0
                s = fma(a, a, s)
        p.append(s)
    return scale * group_sum_mirrors(p)


def half_step(passfn, regfn, proxfn, x, alpha, nobs, min_stepsize=0.01):
    """One segment's half-step (src/algorithms/proxgrad.jl:118-156): passfn(x, grad) -> (J, g); returns (x, alpha, J, trials)."""
    k = len(x)
    Jold, g = passfn(x, True)
    Jold += regfn(x)
    l = float(nobs) + 1.0
    trials = 0
    while alpha > min_stepsize:
        s = alpha / l
        xn = proxfn([fma(-s, g[c], x[c]) for c in range(k)], s)
        Jn, _ = passfn(xn, False)
        Jn += regfn(xn)
        trials += 1
        if Jn < Jold:
            return xn, alpha * 1.05, Jn, trials
        alpha *= .7
        if alpha < min_stepsize:
            alpha = min_stepsize * 1.1
            break
    return list(x), alpha, Jold, trials


def lane_kernel_pass(idx, vals, xv, fac, k, TILE, tiles_per_sup, n_other, scale, grad, gseg, KP=32):
    """csrc/glrm_lane.hpp: lane_pass_kernel, literally.  ONE lane owns the segment; register i holds the 16-byte chunk i ^ p of x, g and y,
    p = gseg & 15; the dot product is two fma chains -- uA over the even registers, uB over the odd ones -- added once; the loss terms go to J0 /
    J1 by the entry's position inside its tile window modulo 2 and are added at the end of the pass (one pass = one super-tile, or all tiles
    when tiles_per_sup = 0); gradient terms in list order; the super-tiles' partial sums are added in order from 0 (col_reduce_kernel)."""
    C = KP // 2
    p = gseg & (C - 1)

    def reg(v, i):  # the double2 register i of a vector: chunk i ^ p (zero padded beyond k)
        c = (i ^ p) * 2
        return (v[c] if c < k else 0.0, v[c + 1] if c + 1 < k else 0.0)

    ntiles = (n_other + TILE - 1) // TILE
    tps = tiles_per_sup if tiles_per_sup > 0 else ntiles
    nsup = (ntiles + tps - 1) // tps
    Jtot, gtot = 0.0, [0.0] * k
    pos = 0
    for s in range(nsup):
        J0 = J1 = 0.0
        g = [[0.0, 0.0] for _ in range(C)]
        for t in range(s * tps, min((s + 1) * tps, ntiles)):
            hi = min((t + 1) * TILE, n_other)
            e = 0
            while pos < len(idx) and idx[pos] < hi:
                y = fac[idx[pos]]
                uA = uB = 0.0
                for i in range(0, C, 2):
                    xa, ya = reg(xv, i), reg(y, i)
                    uA = fma(xa[0], ya[0], uA)
                    uA = fma(xa[1], ya[1], uA)
                    xb, yb = reg(xv, i + 1), reg(y, i + 1)
                    uB = fma(xb[0], yb[0], uB)
                    uB = fma(xb[1], yb[1], uB)
                dot = uA + uB
                L, dL = quad_loss(scale, dot, vals[pos])
                if e & 1:
                    J1 += L
                else:
                    J0 += L
                if grad:
                    for i in range(C):
                        yi = reg(y, i)
                        g[i][0] = fma(dL, yi[0], g[i][0])
                        g[i][1] = fma(dL, yi[1], g[i][1])
                pos += 1
                e += 1
        Js = J0 + J1
        gs = [0.0] * k
        for i in range(C):
            c = (i ^ p) * 2
            for h in (0, 1):
                if c + h < k:
                    gs[c + h] = g[i][h]
        if tiles_per_sup > 0:
            Jtot += Js
            gtot = [gtot[c] + gs[c] for c in range(k)]
        else:
            Jtot, gtot = Js, gs
    return Jtot, gtot
