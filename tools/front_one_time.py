#!/usr/bin/env python3
"""Duration of the front end's one-launch form (front_one_kernel) for a symbol-sized call, by events around the launch, against the
five launches (t2gpu_front_set_chain(h, 0)): n samples per call (default 33024), repeated."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from sdr_receiver_dvb_t2_amd import front
n = int(sys.argv[1]) if len(sys.argv) > 1 else 33024
rng = np.random.default_rng(1)
di = torch.from_numpy(rng.integers(-3000, 3000, n, dtype=np.int16)).cuda()
dq = torch.from_numpy(rng.integers(-3000, 3000, n, dtype=np.int16)).cuda()
out = torch.empty(n + 64, dtype=torch.complex64, device="cuda")
for one in (1, 0):
    f = front.front_end(max_samples=1 << 17)
    f._l.t2gpu_front_set_chain(f.h, one)
    ts = []
    for it in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f.execute_dev(di, dq, [n], out, [np.float32(0.01)], [np.float32(1e-5)], [f.resample - 8e-9 * (it % 3)])
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts = sorted(ts[10:])
    print("%s: median %.1f us  min %.1f  p90 %.1f" % ("one launch " if one else "five launches", ts[len(ts) // 2], ts[0], ts[9 * len(ts) // 10]))
    f.close()
