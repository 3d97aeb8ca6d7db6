"""The Julia shim (julia/HipGLRM.jl) cannot run here (no julia binary), so its struct mirrors are checked statically: every
`struct C*` of the shim is parsed and compared, field by field (order, type, offset, total size), with the struct of the same role in
include/glrm_hip.h as the C compiler lays it out.  A drifted mirror would make `ccall` hand the library garbage (glrm_options
had grown by 8 bytes under an unchanged ABI number)."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "julia", "HipGLRM.jl")
HEADER = os.path.join(ROOT, "include", "glrm_hip.h")

PAIRS = {"CLoss": "glrm_loss", "CReg": "glrm_reg", "CProblem": "glrm_problem", "CParams": "glrm_params", "COptions": "glrm_options",
         "CMultiOptions": "glrm_multi_options", "CSignature": "glrm_signature", "CSumOrder": "glrm_sum_order"}
JL_C = {"Int32": ("int32_t", 4), "Int64": ("int64_t", 8), "Float64": ("double", 8), "UInt64": ("uint64_t", 8)}


JULIA_FILES = ("HipGLRM.jl", "HipGLRMDescriptors.jl", "HipGLRMHandle.jl", "HipGLRMExtras.jl")


def julia_structs():
    txt = "\n".join(open(os.path.join(ROOT, "julia", f)).read() for f in JULIA_FILES)
    out = {}
    for m in re.finditer(r"^struct (C\w+)\s*;?(.*?)\bend\b", txt, flags=re.S | re.M):
        body = re.sub(r"#.*", "", m.group(2))
        out[m.group(1)] = re.findall(r"(\w+)::([\w{}]+)", body)
    return out


def c_structs():
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", txt, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(?:const\s+)?([\w ]+?)\s*(\*?)\s*(\w+(?:\s*,\s*\w+)*)$", decl)
            typ = mm.group(1).strip() + ("*" if mm.group(2) else "")
            for name in mm.group(3).split(","):
                fields.append((name.strip(), typ))
        out[m.group(1)] = fields
    return out


def c_layout(structs):
    """offsetof / sizeof from the C compiler itself."""
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "glrm_hip.h"', "int main(void) {"]
    for s in structs:
        src.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for name, _ in structs[s]:
            src.append(f'printf("{s}.{name} %zu\\n", offsetof({s}, {name}));')
    src.append("return 0; }")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "l.c")
        open(c, "w").write("\n".join(src))
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", os.path.join(d, "l")], check=True)
        out = subprocess.run([os.path.join(d, "l")], check=True, capture_output=True, text=True).stdout
    return {ln.split()[0]: int(ln.split()[1]) for ln in out.splitlines()}


def julia_layout(fields):
    """Julia lays an isbits struct out like C: every field aligned to its size, total padded to the largest alignment."""
    off, res, big = 0, [], 1
    for name, typ in fields:
        size = 8 if typ.startswith("Ptr{") else JL_C[typ][1]
        off = (off + size - 1) // size * size
        res.append((name, off))
        off += size
        big = max(big, size)
    return res, (off + big - 1) // big * big


def test_every_struct_mirror_matches_the_header():
    js, cs = julia_structs(), c_structs()
    assert set(PAIRS) <= set(js), sorted(set(PAIRS) - set(js))
    lay = c_layout({c: cs[c] for c in PAIRS.values()})
    for jname, cname in PAIRS.items():
        jf, cf = js[jname], cs[cname]
        assert [n for n, _ in jf] == [n for n, _ in cf], (jname, jf, cf)          # same fields in the same order
        for (n, jt), (_, ct) in zip(jf, cf):
            if ct.endswith("*"):
                assert jt.startswith("Ptr{"), (jname, n, jt, ct)
                inner = jt[4:-1]
                want = {"CLoss": "glrm_loss", "CReg": "glrm_reg", "Cvoid": "void"}.get(inner) or JL_C[inner][0]
                assert ct[:-1].strip() == want, (jname, n, jt, ct)
            else:
                assert JL_C[jt][0] == ct, (jname, n, jt, ct)
        offs, size = julia_layout(jf)
        assert size == lay[cname], (jname, size, lay[cname])
        for n, o in offs:
            assert o == lay[f"{cname}.{n}"], (jname, n, o, lay[f"{cname}.{n}"])


def test_shim_checks_the_abi_version_of_the_header():
    txt = open(SHIM).read()
    v = int(re.search(r"#define GLRM_HIP_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert int(re.search(r"const ABI_VERSION = (\d+)", txt).group(1)) == v
    assert "check_abi()" in txt and ":glrm_hip_version" in txt


def test_shim_prints_like_the_reference():
    """src/algorithms/proxgrad.jl:210-216: 'Iteration i' is printed for every 10th iteration the loop did not break at -- the last one
    (max_iter) included."""
    txt = open(SHIM).read()
    assert 'stopped || println("Iteration $it: objective value = $(obj[i])")' in txt
    assert "i < nrec[]" not in txt


def test_every_ccall_names_a_declared_symbol():
    declared = set(re.findall(r"\b(glrm_hip_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)))
    for f in JULIA_FILES:
        used = set(re.findall(r":(glrm_hip_\w+)", open(os.path.join(ROOT, "julia", f)).read()))
        assert used <= declared, (f, sorted(used - declared))
        assert used or f == "HipGLRMDescriptors.jl", f


def test_every_ccall_target_is_a_literal():
    """Julia resolves a `ccall` target at compile time: `(:symbol, LIB)` must be written out in the call (LIB is a `const`), a target
    chosen by an expression -- `ccall(multi ? (:a, LIB) : (:b, LIB), ...)` -- is a TypeError at run time.  Nothing here can run Julia, so
    the shape is checked textually."""
    for f in JULIA_FILES:
        txt = re.sub(r"#.*", "", open(os.path.join(ROOT, "julia", f)).read())
        calls = len(re.findall(r"\bccall\(", txt))
        literal = len(re.findall(r"\bccall\(\s*\(:glrm_hip_\w+, LIB\)\s*,", txt))
        assert calls == literal, (f, calls, literal)


def test_the_marshalling_core_stays_within_its_line_budget():
    """the Julia side is <= 150 lines of pure marshalling.  Counted: the code lines (not blank, not comment) of
    julia/HipGLRM.jl -- struct mirrors, the params type, Omega -> CSR / CSC, create, fit!; the type -> descriptor tables, the handle cache
    and the entry points around fit! live in the files it includes."""
    code = [ln for ln in open(SHIM).read().splitlines() if ln.strip() and not ln.strip().startswith("#")]
    assert len(code) <= 150, len(code)
    for f in ("HipGLRMDescriptors.jl", "HipGLRMHandle.jl"):
        assert f'include("{f}")' in open(SHIM).read()


def test_a_sparse_matrix_is_handed_over_without_a_lookup_per_entry():
    """the shim flattened Omega through a per-entry closure A[e, j] -- a binary search per observation on a
    SparseMatrixCSC, twice, for 1e9 observations.  The sparse constructor's lists ARE the CSC arrays (src/glrm.jl:46-48): the column view
    is taken from colptr / rowval / nzval, the row view is left to the engine (GLRM_PROBLEM_ROWS_FROM_COLS, derived on the device), and
    A[e, j] is only reached for lists the caller built."""
    txt = open(SHIM).read()
    body = txt[txt.index("function cols_from_csc"):txt.index("function omega_views")]
    assert "A.colptr" in body and "A.rowval" in body and "A.nzval" in body and "A[" not in body
    ov = txt[txt.index("function omega_views"):txt.index("lasterr()")]
    assert ov.index("cols_from_csc") < ov.index("rows()...") and "Int32(8)" in ov          # the lookup path is the fallback
    assert "csc_is_omega(A, glrm.observed_examples)" in ov and "rows_are_transpose(A, glrm.observed_features)" in ov
    flag = int(re.search(r"#define GLRM_PROBLEM_ROWS_FROM_COLS (\d+)", open(HEADER).read()).group(1))
    assert flag == 8 and "CProblem(m, n, glrm.k, flags," in txt
