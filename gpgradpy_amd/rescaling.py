"""Host-side data rescaling for the well-conditioning methods 'rescale_origin', 'rescale_eta_vary', 'dflt_vmin' and
'dflt_vmax' -- the interface of the reference's `Rescaling` (base/Rescaling.py:395-557 and its four mix-ins, :20-393),
restated.  Nothing here touches the device: the GaussianProcess pushes the SCALED x / f / grad to the GPU and scales
the posterior's outputs back.

Parameter space:   x_scl = (x_init - x_shift) * xvec_scale, where xvec_scale = xvec_scale_in * c and the scalar c makes the
                   minimum ('set_vmin') or maximum ('set_vmax') pairwise Euclidean distance of the scaled points equal
                   to `dist_set` (Rescaling.py:72-125).
Objective:         obj_scl = (obj - obj_shift) * obj_scale, gradients * obj_scale / xvec_scale, Hessians * obj_scale / xvec_scale^2
                   (Rescaling.py:134-183); obj_scale maps the range of the data on `rangeobj_max_dflt` ('dflt_max', :199-216).
Only what the Gaussian process itself uses is here; the reference class also carries constraint data, box and linear constraints
(:223-393) for the Bayesian optimiser that sits above the package -- host bookkeeping outside this path.

The Hessian factors broadcast over the LAST axis only -- entry (i, j) is scaled by xvec_scale[j]^2, not xvec_scale[i] *
xvec_scale[j] -- exactly as the reference's arrays of shape [1, dim] do (Rescaling.py:147,173); the two agree whenever the
scaling is isotropic, which it is until the 'rescale' optimisation loop has run.
"""
import numpy as np
from scipy.spatial.distance import cdist


def calc_dist_min(Xmat):
    """Smallest distance between two rows of Xmat (CommonFun.py:16-34); NaN for a single point."""
    n = Xmat.shape[0]
    if n == 1:
        return np.nan
    D = cdist(Xmat, Xmat, 'euclidean')
    D[np.diag_indices(n)] = np.nan
    return np.nanmin(D)


def calc_dist_max(Xmat):
    """Largest distance between two rows of Xmat (CommonFun.py:36-54); NaN for a single point."""
    if Xmat.shape[0] == 1:
        return np.nan
    return np.nanmax(cdist(Xmat, Xmat, 'euclidean'))


def _opt(a, factor, shift=None):
    if a is None:
        return None
    return a * factor if shift is None else (a - shift) * factor


class Rescaling:
    tol_min_range_obj = 1e-20
    tol_min_dist_x = 1e-14

    x_scl_method_avail = ['set_vmin', 'set_vmax', None]
    obj_scl_method_avail = ['dflt_max', None]

    vmin_dflt = 1
    vmax_dflt = 1
    rangeobj_max_dflt = 100

    _xdata_set = _obj_data_set = False
    use_x_shift = use_obj_shift = True
    x_scl_method = obj_scl_method = None
    dist_set = None
    x_shift = xvec_scale = obj_shift = obj_scale = np.nan
    on_change = None              # callable run after the scaled data changed (the GaussianProcess re-sends them to the device)

    def __init__(self, x_init, idx_xbest=None, use_x_shift=True, x_scl_method=None, dist_set=None):
        assert x_init.ndim == 2, f'x_init needs to be 2D but it has the shape {x_init.shape}'
        assert x_scl_method in self.x_scl_method_avail, f'Requested x_scl_method = {x_scl_method} is not available'
        self._xdata_set = True
        self.n_eval, self.dim = x_init.shape
        self.x_init = x_init
        self.idx_xbest = self.n_eval - 1 if idx_xbest is None else idx_xbest       # Rescaling.py:477-480
        self.use_x_shift = use_x_shift
        self.x_scl_method = x_scl_method
        self.dist_set = dist_set
        self.set_xscale_data()

    # ---- parameter space -------------------------------------------------------------------------------------
    def x_init_2_scl(self, x_init):
        if x_init.ndim not in (1, 2):
            raise Exception(f'Shape of x_init = {x_init.shape}')
        return (x_init - self.x_shift) * self.xvec_scale            # broadcasting covers [dim] and [nx, dim]

    def x_scl_2_init(self, x_scl):
        if x_scl.ndim not in (1, 2):
            raise Exception(f'Shape of x_scl = {x_scl.shape}')
        return x_scl / self.xvec_scale + self.x_shift

    def dist_init_2_scl(self, dist_init):
        return dist_init * np.mean(self.xvec_scale)

    def dist_scl_2_init(self, dist_scl):
        return dist_scl / np.mean(self.xvec_scale)

    def _calc_x_shift_n_scale(self, x_shift_in=None, xvec_scale_in=None):
        if x_shift_in is None:
            x_shift = self.x_init[self.idx_xbest, :] if self.use_x_shift else np.zeros(self.dim)
        else:
            x_shift_in = np.atleast_1d(x_shift_in)
            assert x_shift_in.size == self.dim, f'Wrong size for x_shift_in: {x_shift_in.shape}'
            assert x_shift_in.ndim <= 1, f'Wrong shape for x_shift_in: {x_shift_in.shape}'
            x_shift = 1 * x_shift_in
        if xvec_scale_in is None:
            xvec_scale_in = np.ones(self.dim)
        else:
            assert xvec_scale_in.size == self.dim, f'Wrong shape for xvec_scale_in: {xvec_scale_in.shape}'
            assert np.all(xvec_scale_in > 0), f'All entries must be positive: {xvec_scale_in}'
        x_v1 = (self.x_init - x_shift[None, :]) * xvec_scale_in[None, :]
        if self.n_eval == 1 or self.x_scl_method is None:
            coeff = 1
        elif self.x_scl_method == 'set_vmin':
            target = self.dist_set if self.dist_set is not None else self.vmin_dflt
            coeff = target / np.max((self.tol_min_dist_x, calc_dist_min(x_v1)))
        elif self.x_scl_method == 'set_vmax':
            target = self.dist_set if self.dist_set is not None else self.vmax_dflt
            coeff = target / calc_dist_max(x_v1)
        else:
            raise Exception(f'Method of x_scl_method = {self.x_scl_method} is unavailable')
        return x_shift, xvec_scale_in * coeff

    def set_xscale_data(self, x_shift_in=None, xvec_scale_in=None):
        """New shift / anisotropic scale of the parameter space; everything that depends on it is rescaled (Rescaling.py:44-70)."""
        self.x_shift, self.xvec_scale = self._calc_x_shift_n_scale(x_shift_in, xvec_scale_in)
        self.x_scl = self.x_init_2_scl(self.x_init)
        self._Rtensor_scl = None
        if self._obj_data_set:
            self._calc_n_set_scl_obj()
        if self.on_change is not None:
            self.on_change()

    @property
    def Rtensor_scl(self):
        """[dim, n, n] difference tensor of the scaled points (Rescaling.py:130): built when somebody reads it -- the device
        path never does."""
        if self._Rtensor_scl is None:
            self._Rtensor_scl = np.ascontiguousarray(self.x_scl.T[:, :, None] - self.x_scl.T[:, None, :])
        return self._Rtensor_scl

    def get_init_x(self):
        return self.x_init

    def get_scl_x(self):
        return self.x_scl

    def get_scl_x_w_dist(self):
        return self.x_scl, self.Rtensor_scl

    # ---- objective data ------------------------------------------------------------------------------------------
    def _derivative_factors(self, to_scl, out_scale):
        s = self.xvec_scale[None, :]
        if to_scl:
            return out_scale / s, out_scale / s ** 2
        return s / out_scale, s ** 2 / out_scale

    def obj_init_2_scl(self, mu_in=None, sig_in=None, dmudx_in=None, dsigdx_in=None, d2mudx2_in=None, d2sigdx2_in=None):
        assert self._obj_data_set, 'Must call set_obj_data prior to this method'
        fg, fh = self._derivative_factors(True, self.obj_scale)
        return (_opt(mu_in, self.obj_scale, self.obj_shift), _opt(sig_in, self.obj_scale), _opt(dmudx_in, fg),
                _opt(dsigdx_in, fg), _opt(d2mudx2_in, fh), _opt(d2sigdx2_in, fh))

    def obj_scl_2_init(self, mu_scl=None, sig_scl=None, dmudx_scl=None, dsigdx_scl=None, d2mudx2_scl=None, d2sigdx2_scl=None):
        assert self._obj_data_set, 'Must call set_obj_data prior to this method'
        fg, fh = self._derivative_factors(False, self.obj_scale)
        return (None if mu_scl is None else mu_scl / self.obj_scale + self.obj_shift,
                None if sig_scl is None else sig_scl / self.obj_scale, _opt(dmudx_scl, fg), _opt(dsigdx_scl, fg),
                _opt(d2mudx2_scl, fh), _opt(d2sigdx2_scl, fh))

    def _calc_obj_scaling(self, obj_init):
        obj_shift = obj_init[self.idx_xbest] if self.use_obj_shift else 0
        if obj_init.size == 1 or self.obj_scl_method is None:
            return obj_shift, 1
        if self.obj_scl_method == 'dflt_max':
            spread = np.max((self.tol_min_range_obj, np.max(obj_init) - np.min(obj_init)))
            return obj_shift, self.rangeobj_max_dflt / spread
        raise Exception(f'Unavailable method of obj_scl_method = {self.obj_scl_method}')

    def _calc_n_set_scl_obj(self):
        self.obj_scl, self.std_obj_scl, self.grad_scl, self.std_grad_scl = \
            self.obj_init_2_scl(self.obj_init, self.std_obj_init, self.grad_init, self.std_grad_init)[:4]

    def set_obj_scaling(self, obj_shift=None, obj_scale=None):
        assert self._obj_data_set, 'Must call set_obj_data prior to this method'
        if obj_shift is not None:
            self.obj_shift = obj_shift
        if obj_scale is not None:
            self.obj_scale = obj_scale
        self._calc_n_set_scl_obj()
        if self.on_change is not None:
            self.on_change()

    def set_obj_data(self, obj_init, std_obj_init, grad_init, std_grad_init, use_obj_shift=True, obj_scl_method='dflt_max'):
        assert self._xdata_set, 'Must call the method set_xdata prior to set_obj_data'
        assert obj_scl_method in self.obj_scl_method_avail, f'Requested obj_scl_method = {obj_scl_method} is not available'
        assert self.n_eval == obj_init.size, \
            f'Dimension of n_eval = {self.n_eval} do not match with size of obj_init: {obj_init.size}'
        self._obj_data_set = True
        self.obj_init, self.std_obj_init = obj_init, std_obj_init
        self.grad_init, self.std_grad_init = grad_init, std_grad_init
        self.obj_scl_method, self.use_obj_shift = obj_scl_method, use_obj_shift
        self.set_obj_scaling(*self._calc_obj_scaling(obj_init))

    def get_init_obj_data(self):
        assert self._obj_data_set, 'Must call the method set_obj_data prior to this method'
        return self.obj_init, self.std_obj_init, self.grad_init, self.std_grad_init

    def get_scl_obj_data(self):
        assert self._obj_data_set, 'Must call the method set_obj_data prior to this method'
        return self.obj_scl, self.std_obj_scl, self.grad_scl, self.std_grad_scl
