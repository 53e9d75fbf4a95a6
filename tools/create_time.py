"""Diagnostic: where set_data spends its time (context creation / destruction through the C ABI, host-side scaling)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import gpgradpy_amd
from gpgradpy_amd import _lib
lib = _lib.load()


for n, d in [(25, 4), (300, 4)]:
    t = {'create': [], 'destroy': []}
    ctx = C.c_void_p()
    for _ in range(8):
        t0 = time.perf_counter()
        rc = lib.gpg_create(C.byref(ctx), 0, n, d, 1, _lib.GPG_KERNEL['SqExp'])
        t1 = time.perf_counter()
        assert rc == 0
        lib.gpg_destroy(ctx)
        t2 = time.perf_counter()
        t['create'].append((t1 - t0) * 1e3); t['destroy'].append((t2 - t1) * 1e3)
    print(n, d, 'create ms', ' '.join('%.2f' % v for v in t['create']), '| destroy ms', ' '.join('%.2f' % v for v in t['destroy']))
