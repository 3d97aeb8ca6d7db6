"""Flat binary twins of the golden fixtures, readable from Julia without any package (julia/crosscheck.jl), and the reader of the
dumps that script writes after running the REAL LowRankModels.fit! on the same inputs.

    python tests/golden/fixture_bin.py            # (re)writes tests/golden/bin/<name>.bin from tests/golden/<name>.npz

<name>.bin (little endian):
    char[8]   "GLRMFIX1"
    int64[10] m, n, k, d, nnz_r, nnz_c, n_losses (= n), n_rx (= m), n_ry (= n), nobj
    f64[7]    stepsize, max_iter, inner_iter_X, inner_iter_Y, abs_tol, rel_tol, min_stepsize       (ProxGradParams)
    int64[m+1] rowptr; int32[nnz_r] colidx (0-based); f64[nnz_r] rowvals                            (observed_features, list order)
    int64[n+1] colptr; int32[nnz_c] rowidx (0-based); f64[nnz_c] colvals                            (observed_examples)
    f64[n][5] losses (kind, dim, scale, p0, p1 -- include/glrm_hip.h); f64[m][3] rx (kind, wrap, scale); f64[n][3] ry
    f64[k*m]  X0 (column-major k x m); f64[k*d] Y0
    f64[nobj] objective; f64[k*m] X; f64[k*d] Y                                                     (the oracle's result)

<name>.ref.bin, written by julia/crosscheck.jl (reference run on the same inputs):
    char[8]   "GLRMREF1"
    int64[4]  nobj, k, m, d
    f64[nobj] ch.objective; f64[nobj] ch.times; f64[k*m] X; f64[k*d] Y
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin")
REF = os.path.join(HERE, "ref")


def _full(desc, count, width):
    d = np.asarray(desc, dtype=np.float64).reshape(-1, width)
    return np.repeat(d, count, axis=0) if len(d) == 1 and count > 1 else d


def write_bin(npz_path, out_path):
    z = np.load(npz_path)
    m, n, k = int(z["m"]), int(z["n"]), int(z["k"])
    X0, Y0 = np.asfortranarray(z["X0"]), np.asfortranarray(z["Y0"])
    d = Y0.shape[1]
    losses, rx, ry = _full(z["losses"], n, 5), _full(z["rx"], m, 3), _full(z["ry"], n, 3)
    obj = np.asarray(z["objective"], dtype=np.float64)
    with open(out_path, "wb") as f:
        f.write(b"GLRMFIX1")
        f.write(struct.pack("<10q", m, n, k, d, len(z["colidx"]), len(z["rowidx"]), n, m, n, len(obj)))
        f.write(np.asarray(z["params"], dtype="<f8").tobytes())
        for name, dt in (("rowptr", "<i8"), ("colidx", "<i4"), ("rowvals", "<f8"), ("colptr", "<i8"), ("rowidx", "<i4"), ("colvals", "<f8")):
            f.write(np.asarray(z[name], dtype=dt).tobytes())
        for a in (losses, rx, ry):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        f.write(X0.tobytes(order="F")); f.write(Y0.tobytes(order="F"))
        f.write(obj.astype("<f8").tobytes())
        f.write(np.asfortranarray(z["X"]).tobytes(order="F")); f.write(np.asfortranarray(z["Y"]).tobytes(order="F"))


def read_bin(path):
    """-> dict with the arrays of a <name>.bin (used by the tests to check the twin against the npz)."""
    b = open(path, "rb").read()
    assert b[:8] == b"GLRMFIX1"
    m, n, k, d, nzr, nzc, nl, nrx, nry, nobj = struct.unpack_from("<10q", b, 8)
    off = 8 + 80
    out = dict(m=m, n=n, k=k, d=d)

    def take(name, dt, count, shape=None, order="C"):
        nonlocal off
        a = np.frombuffer(b, dtype=dt, count=count, offset=off)
        off += a.nbytes
        out[name] = a.reshape(shape, order=order) if shape else a

    take("params", "<f8", 7)
    take("rowptr", "<i8", m + 1); take("colidx", "<i4", nzr); take("rowvals", "<f8", nzr)
    take("colptr", "<i8", n + 1); take("rowidx", "<i4", nzc); take("colvals", "<f8", nzc)
    take("losses", "<f8", nl * 5, (nl, 5)); take("rx", "<f8", nrx * 3, (nrx, 3)); take("ry", "<f8", nry * 3, (nry, 3))
    take("X0", "<f8", k * m, (k, m), "F"); take("Y0", "<f8", k * d, (k, d), "F")
    take("objective", "<f8", nobj); take("X", "<f8", k * m, (k, m), "F"); take("Y", "<f8", k * d, (k, d), "F")
    assert off == len(b), (off, len(b))
    return out


def read_ref(path):
    """-> (objective, times, X, Y) of a dump written by julia/crosscheck.jl."""
    b = open(path, "rb").read()
    assert b[:8] == b"GLRMREF1", "not a crosscheck.jl dump"
    nobj, k, m, d = struct.unpack_from("<4q", b, 8)
    off = 8 + 32
    obj = np.frombuffer(b, "<f8", nobj, off); off += 8 * nobj
    tim = np.frombuffer(b, "<f8", nobj, off); off += 8 * nobj
    X = np.frombuffer(b, "<f8", k * m, off).reshape((k, m), order="F"); off += 8 * k * m
    Y = np.frombuffer(b, "<f8", k * d, off).reshape((k, d), order="F"); off += 8 * k * d
    assert off == len(b)
    return obj, tim, X, Y


if __name__ == "__main__":
    os.makedirs(BIN, exist_ok=True)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            out = os.path.join(BIN, fn[:-4] + ".bin")
            write_bin(os.path.join(HERE, fn), out)
            print(f"{out}: {os.path.getsize(out) / 1024:.0f} kB")
