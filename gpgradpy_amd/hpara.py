"""Data contracts of the hot path (same field names as the reference)."""
from dataclasses import dataclass

import numpy as np


@dataclass
class HparaOptzVal:
    """Hyperparameter container -- reference GpHpara.py:12-19."""
    beta: np.ndarray = None
    theta: np.ndarray = None
    kernel: float = None
    varK: float = None
    var_fval: float = None
    var_fgrad: float = None


@dataclass(frozen=True)
class HparaOptzInfo:
    """Layout of the numerically optimised hyperparameter vector -- reference GpHparaOptz.py:18-31."""
    n_hp: int = None
    has_theta: bool = False
    idx_theta: np.ndarray = None
    has_kernel: bool = False
    idx_kernel: np.ndarray = None
    has_varK: bool = False
    idx_varK: np.ndarray = None
    has_var_fval: bool = False
    idx_var_fval: np.ndarray = None
    has_var_fgrad: bool = False
    idx_var_fgrad: np.ndarray = None
    bvec_log_optz: np.ndarray = None


@dataclass
class LkdInfo:
    """Result of one likelihood evaluation -- reference CalcLkd.py:14-26."""
    hp_beta: np.ndarray = None
    hp_beta_grad: np.ndarray = None
    hp_varK: float = None
    hp_varK_grad: np.ndarray = None
    ln_det_Kmat: float = None
    ln_det_Kmat_grad: np.ndarray = None
    ln_lkd: float = None
    ln_lkd_grad: np.ndarray = None
    cond: float = None
    cond_grad: np.ndarray = None
    data_vec: np.ndarray = None
