"""Host-side conveniences of the reference that are OUT OF SCOPE for the product but that the reference's own test
scripts use to build models (test/hello_world.jl, test/runtests.jl): kept on the test side only."""
from .fit_dataframe import glrm_from_dataframe, probabilistic_losses, robust_losses
from .prob_scale import prob_scale_
from .scaling import M_estimator, avgerror, equilibrate_variance_
from .simple_glrms import kmeans, nnmf, pca, qpca, rpca
