"""gpgradpy_amd: MI355X-native hot path of the gradient-enhanced GP package GpGradPy.

Host-side mirror of the reference's `GaussianProcess` surface for ONE path -- kernel-matrix build,
preconditioner + nugget, Cholesky, marginal log-likelihood, posterior mean / std -- running on
hand-written HIP kernels behind the C ABI of include/gpgrad.h.  See DESIGN.md.
"""
from .hpara import HparaOptzVal, HparaOptzInfo, LkdInfo
from .gaussian_process import GaussianProcess
from .multistart import select_best_restart, shard_rows

__all__ = ["GaussianProcess", "HparaOptzVal", "HparaOptzInfo", "LkdInfo", "select_best_restart", "shard_rows"]
