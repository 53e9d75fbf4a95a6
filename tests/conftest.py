import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(path):
    z = np.load(path, allow_pickle=False)
    c = {k: z[k] for k in z.files}
    for k in ("name", "kernel", "noise", "wellcond"):
        if k in c:
            c[k] = str(c[k])
    for k in ("n", "d", "n_data"):
        if k in c:
            c[k] = int(c[k])
    for k in ("use_grad", "b_has_noisy_data", "b_chofac_good", "chofac_lower"):
        if k in c:
            c[k] = bool(c[k])
    for k in ("varK_in", "var_fval", "var_fgrad", "etaK", "hp_varK", "ln_det_Kmat", "ln_lkd", "varK_model"):
        if k in c:
            c[k] = float(c[k])
    if "bvec_use_grad" in c:
        c["bvec_use_grad"] = c["bvec_use_grad"].astype(bool)
    # what the oracle takes as `kernel`: the name, or ("RatQu", alpha) for the kernel with a hyperparameter of its own
    c["kernel_o"] = (c["kernel"], float(c["hp_kernel"])) if c.get("kernel") == "RatQu" else c.get("kernel")
    c["std_f"] = None if c["std_f"].size == 0 else c["std_f"]
    c["std_g"] = None if c["std_g"].size == 0 else c["std_g"]
    return c


def golden_case_paths():
    paths = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [p for p in paths if not os.path.basename(p).startswith(("multistart_", "nugget_", "optz_", "hess_", "cond_", "evar_", "failobj_", "rescale_", "direct_", "precon_", "kgrad_"))]


def case_id(path):
    return os.path.splitext(os.path.basename(path))[0]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
