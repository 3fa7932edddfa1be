#!/usr/bin/env python3
"""What a SHORT kernel costs while LDPC decodes run beside it on other streams (the slot-shaped path's per-symbol launches next to the
decodes of the frame before). A chain of tiny launches (a 4 KB element-wise add each, one stream), timed by events around every
launch, alone and with n one-batch decodes (16 workgroups each, plain launches, streams at the lowest priority) in flight."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
l = pkg.lib()
N = int(os.environ.get("N", "7"))
decs = [pkg.ldpc_decoder(1, 3, max_frames=32) for _ in range(N)]
for d in decs:
    l.t2gpu_ldpc_set_plain_launch(d._h, 1)
llr = torch.from_numpy(np.random.default_rng(1).integers(-20, 21, size=(32, 64800), dtype=np.int8)).cuda()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
streams = [torch.cuda.Stream(priority=0) for _ in range(N)]
chain = torch.cuda.Stream(priority=-1)
x = torch.zeros(1024, device="cuda")
big = torch.zeros(1 << 22, device="cuda")


def run_chain(n=60, t=x):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    with torch.cuda.stream(chain):
        for a, b in ev:
            a.record(chain)
            t.add_(1.0)
            b.record(chain)
    chain.synchronize()
    d = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return d[len(d) // 2], d[-1], sum(d) / len(d)


torch.cuda.synchronize()
for tname, t in (("4 KB add", x), ("16 MB add", big)):
    print("%s alone:                  median %.1f us  max %.1f  mean %.1f" % ((tname,) + run_chain(t=t)))
    for cnt in (1, 4, N):
        for d, s in zip(decs[:cnt], streams[:cnt]):
            with torch.cuda.stream(s):
                d.execute_dev(llr)
        time.sleep(0.0003)
        r = run_chain(t=t)
        torch.cuda.synchronize()
        print("%s beside %d decodes:       median %.1f us  max %.1f  mean %.1f" % ((tname, cnt) + r))
