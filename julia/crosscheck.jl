# crosscheck.jl -- pins the CPU oracle (and through it the HIP engine) to the REAL reference: runs LowRankModels.jl's own
# fit!(glrm, ProxGradParams(...)) (src/algorithms/proxgrad.jl:34-220) on the inputs of a committed golden fixture and dumps
# ch.objective, ch.times, X and Y in the flat format tests/golden/fixture_bin.py documents.
#
#   julia julia/crosscheck.jl /path/to/LowRankModels.jl tests/golden/bin tests/golden/ref        # every <name>.bin of the directory
#   julia julia/crosscheck.jl /path/to/LowRankModels.jl tests/golden/bin/c4.bin tests/golden/ref
#
# then   python -m pytest tests/test_golden.py -k reference_dump     compares every tests/golden/ref/<name>.ref.bin with the oracle's
# trajectory stored in the fixture (and, on the GPU box, tests/test_gpu_parity.py::test_golden_fixtures compares the engine with the
# same numbers).  NOT EXECUTED IN THIS REPOSITORY: the build image and the GPU box have no julia binary (SURVEY.md F2); until a dump
# exists the trajectory parity of the oracle is pinned at operator level only (DESIGN.md section 3).
#
# The script needs nothing but the reference package and its own dependencies; fixtures are read with plain `read(io, T)` calls.
length(ARGS) >= 3 || error("usage: julia crosscheck.jl <LowRankModels.jl checkout> <fixture.bin | directory> <output directory>")
pushfirst!(LOAD_PATH, joinpath(ARGS[1], "src")); pushfirst!(LOAD_PATH, ARGS[1])
using LowRankModels

readvec(io, T, n) = (v = Vector{T}(undef, n); read!(io, v); v)

# glrm_loss / glrm_reg descriptors (include/glrm_hip.h) -> the reference's constructors (src/losses.jl, src/regularizers.jl)
function mkloss(d)
    kind, dim, s, p0, p1 = Int(d[1]), Int(d[2]), d[3], d[4], d[5]
    bin(k, sc) = k == 7 ? LogisticLoss(sc) : HingeLoss(sc)
    kind == 0 ? QuadLoss(s) : kind == 1 ? L1Loss(s) : kind == 2 ? HuberLoss(s; crossover=p0) :
    kind == 3 ? QuantileLoss(s; quantile=p0) : kind == 4 ? PeriodicLoss(p0, s) :
    kind == 5 ? (l = PoissonLoss(); l.scale = s; l) : kind == 6 ? OrdinalHingeLoss(Int(p0), Int(p1), s) :
    kind == 7 ? LogisticLoss(s) : kind == 8 ? WeightedHingeLoss(s; case_weight_ratio=p0) :
    kind == 9 ? MultinomialLoss(dim, s) : kind == 10 ? OvALoss(dim, s; bin_loss=bin(Int(p1), p0)) :
    kind == 11 ? BvSLoss(dim + 1, s; bin_loss=bin(Int(p1), p0)) : kind == 12 ? OrdisticLoss(dim, s) :
    kind == 13 ? MultinomialOrdinalLoss(dim + 1, s) : error("loss kind $kind")
end
function mkreg(d)
    kind, wrap, s = Int(d[1]), Int(d[2]), d[3]
    base = kind == 0 ? ZeroReg() : kind == 1 ? QuadReg(s) : kind == 2 ? OneReg(s) : kind == 3 ? NonNegConstraint() :
           kind == 4 ? UnitOneSparseConstraint() : error("regularizer kind $kind")
    wrap == 0 ? base : wrap == 1 ? lastentry1(base) : wrap == 2 ? lastentry_unpenalized(base) : wrap == 4 ? OrdinalReg(base) :
    wrap == 8 ? MNLOrdinalReg(base) : error("wrap $wrap")
end

function crosscheck(path, outdir)
    io = open(path, "r")
    String(read(io, 8)) == "GLRMFIX1" || error("$path is not a GLRMFIX1 fixture")
    m, n, k, d, nzr, nzc, nl, nrx, nry, nobj = readvec(io, Int64, 10)
    prm = readvec(io, Float64, 7)
    rowptr = readvec(io, Int64, m + 1); colidx = readvec(io, Int32, nzr); rowvals = readvec(io, Float64, nzr)
    colptr = readvec(io, Int64, n + 1); rowidx = readvec(io, Int32, nzc); colvals = readvec(io, Float64, nzc)
    ld = reshape(readvec(io, Float64, 5nl), 5, nl); rxd = reshape(readvec(io, Float64, 3nrx), 3, nrx); ryd = reshape(readvec(io, Float64, 3nry), 3, nry)
    X0 = reshape(readvec(io, Float64, k * m), k, m); Y0 = reshape(readvec(io, Float64, k * d), k, d)
    obj_oracle = readvec(io, Float64, nobj)
    close(io)
    losses = Loss[mkloss(ld[:, j]) for j in 1:n]
    rx = Regularizer[mkreg(rxd[:, i]) for i in 1:m]; ry = Regularizer[mkreg(ryd[:, j]) for j in 1:n]
    # the two Omega views, each from its own list (order and duplicates kept), 1-based
    feats = [Int[colidx[t] + 1 for t in rowptr[i]+1:rowptr[i+1]] for i in 1:m]
    exs = [Int[rowidx[t] + 1 for t in colptr[j]+1:colptr[j+1]] for j in 1:n]
    # A as the element types the reference dispatches on: Bool for ClassificationLoss columns, Int levels for the categorical /
    # ordinal multi-dimensional losses, Float64 otherwise (src/losses.jl:104-106,360-620)
    conv(j, a) = losses[j] isa LowRankModels.ClassificationLoss ? (a == 1.0) : (Int(ld[1, j]) >= 9 ? Int(a) : a)
    A = Array{Any}(undef, m, n); fill!(A, 0.0)
    for i in 1:m, t in rowptr[i]+1:rowptr[i+1]; j = colidx[t] + 1; A[i, j] = conv(j, rowvals[t]); end
    for j in 1:n, t in colptr[j]+1:colptr[j+1]; i = rowidx[t] + 1; A[i, j] = conv(j, colvals[t]); end
    glrm = GLRM(A, losses, rx, ry, Int(k); X=copy(X0), Y=copy(Y0), observed_features=feats, observed_examples=exs, checknan=false)
    p = ProxGradParams(prm[1]; max_iter=Int(prm[2]), inner_iter_X=Int(prm[3]), inner_iter_Y=Int(prm[4]), abs_tol=prm[5], rel_tol=prm[6],
                       min_stepsize=prm[7])
    X, Y, ch = fit!(glrm, p; verbose=false)
    name = replace(basename(path), ".bin" => "")
    open(joinpath(outdir, name * ".ref.bin"), "w") do o
        write(o, "GLRMREF1"); write(o, Int64[length(ch.objective), k, m, d])
        write(o, Vector{Float64}(ch.objective)); write(o, Vector{Float64}(ch.times)); write(o, Matrix{Float64}(X)); write(o, Matrix{Float64}(Y))
    end
    nc = min(length(ch.objective), nobj)
    rel = maximum(abs.(ch.objective[1:nc] .- obj_oracle[1:nc]) ./ max.(abs.(obj_oracle[1:nc]), 1e-300))
    println("$name: reference $(length(ch.objective) - 1) iterations (oracle $(nobj - 1)), objective $(ch.objective[1]) -> $(ch.objective[end]); ",
            "max rel. difference to the oracle's trajectory on the common prefix: $rel")
end

mkpath(ARGS[3])
for f in (isdir(ARGS[2]) ? sort(filter(x -> endswith(x, ".bin"), readdir(ARGS[2]; join=true))) : [ARGS[2]])
    crosscheck(f, ARGS[3])
end
