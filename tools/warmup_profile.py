#!/usr/bin/env python3
"""R&D: per-step wall time of the first steps after start-up (how much
warm-up the bench needs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ludwig_amd
from ludwig_amd import synthetic

lb = ludwig_amd.LB(19, (256, 256, 256), 1, mode=ludwig_amd.FUSED)
lb.relaxation_set("m10", 0.1, 0.3)
m = ludwig_amd.model(19)
synthetic.fill_device(lb, m["cv"], m["wv"], (256, 256, 256))
hy = ludwig_amd.Hydro(lb.nall, lb.device, force=np.zeros((3,) + lb.nall))
torch.cuda.synchronize()
ts = []
for n in range(16):
    t0 = time.perf_counter()
    lb.run(hy, 1)
    lb.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print("ms per step, steps 1..16:", " ".join("%.3f" % t for t in ts))
