#!/usr/bin/env python3
"""Do decodes of one SIMD batch each (16 workgroups of the LDPC kernel), enqueued on streams of their own, run side by side?
t2gpu_ldpc_execute_dev on n handles x n streams with device-resident LLRs (no copies in the way), plain launches; prints the wall time
of n decodes against one."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
l = pkg.lib()
N = 8
decs = [pkg.ldpc_decoder(1, 3, max_frames=32) for _ in range(N)]
for d in decs:
    l.t2gpu_ldpc_set_plain_launch(d._h, int(os.environ.get("PLAIN", "1")))
llr = torch.from_numpy(np.random.default_rng(1).integers(-20, 21, size=(32, 64800), dtype=np.int8)).cuda()
prio = int(os.environ.get("PRIO", "0"))
streams = [torch.cuda.Stream(priority=prio) for _ in range(N)]
torch.cuda.synchronize()
for cnt in (1, 2, 4, 8):
    for rnd in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for d, s in zip(decs[:cnt], streams[:cnt]):
            with torch.cuda.stream(s):
                d.execute_dev(llr)
        torch.cuda.synchronize()
        print("decodes %d round %d: %.2f ms" % (cnt, rnd, (time.perf_counter() - t0) * 1e3))
