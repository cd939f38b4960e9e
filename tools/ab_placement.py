#!/usr/bin/env python3
"""R&D: how much does the placement of the arrays matter inside ONE process?
Allocate the distribution arrays several times (keeping the earlier ones
alive, so every try lives somewhere else), time the same kernel on each."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ludwig_amd
from ludwig_amd import lib as _l
from ludwig_amd import synthetic

nvel, size = 19, (256, 256, 256)
lb = ludwig_amd.LB(nvel, size, 1, mode=ludwig_amd.FUSED)
lb.relaxation_set("m10", 0.1, 0.3)
m = ludwig_amd.model(nvel)
synthetic.fill_device(lb, m["cv"], m["wv"], size)
hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall))
f_init = lb.f.clone()
keep = []
n = nvel * lb.nsite
for k in range(8):
    a = torch.empty((nvel,) + lb.nall, dtype=torch.float64, device=lb.device)
    b = torch.empty((nvel,) + lb.nall, dtype=torch.float64, device=lb.device)
    keep.append((a, b))
    a.copy_(f_init)
    b.zero_()
    torch.cuda.synchronize()
    _l.check(lb._lib.lbmi_lb_bind(lb._h, ctypes.c_void_p(a.data_ptr()),
                                  ctypes.c_void_p(b.data_ptr())))
    lb._a, lb._b = a, b
    lb.run(hy, 6)
    lb.synchronize()
    lb.timing(1)
    lb.run(hy, 40)
    lb.synchronize()
    kms, cnt = lb.timing_read()
    lb.timing(0)
    lb.lb_flush()
    lb.synchronize()
    print("try %d  f @ 0x%x  fprime @ 0x%x  (f mod 1GiB = %4d MiB)  %.4f ms  %.0f GB/s"
          % (k, a.data_ptr(), b.data_ptr(), (a.data_ptr() % (1 << 30)) >> 20, kms / cnt,
             1e-6 * 360 * 256 ** 3 / (kms / cnt)))
