"""ConvergenceHistory / update_ch! (reference: src/convergence.jl:3-27)."""


class ConvergenceHistory:
    def __init__(self, name="unnamed_convergence_history", optval=0):
        self.name = name
        self.objective, self.dual_objective = [], []
        self.primal_residual, self.dual_residual = [], []
        self.times, self.stepsizes = [], []
        self.optval = optval

    def __repr__(self):
        return f"ConvergenceHistory({self.name!r}, {len(self.objective)} entries)"


def update_ch(ch, dt, obj, stepsize=0, pr=0, dr=0):
    """update_ch!(ch, dt, obj): push the objective, accumulate time (src/convergence.jl:16-27)."""
    ch.objective.append(obj)
    ch.primal_residual.append(pr)
    ch.dual_residual.append(dr)
    ch.stepsizes.append(stepsize)
    ch.times.append(dt if not ch.times else ch.times[-1] + dt)
