#!/usr/bin/env python3
"""Per-phase cycle breakdown of the LDPC kernel (wave 0 of every workgroup), via t2gpu_ldpc_profile."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
import oracle_lib as ol
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
group = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rate = int(sys.argv[4]) if len(sys.argv) > 4 else 3           # code rate id of the 64800 code (3 = 3/4, 2 = 2/3 ...)
cid = 6 + rate
if len(sys.argv) > 3 and sys.argv[3] == "noise":       # the bench's load: nothing converges, every batch runs all 25 sweeps
    llr = np.random.default_rng(1).integers(-20, 21, size=(256, 64800), dtype=np.int8)
else:
    info, llr = ol.make_llr(cid, min(frames, 256), 0.60, 1)
llr = np.tile(llr, ((frames + 255) // 256, 1))[:frames]
x = torch.from_numpy(np.ascontiguousarray(llr)).cuda()
dec = pkg.ldpc_decoder(1, rate, max_frames=frames, group=group)
dec.execute_dev(x); torch.cuda.synchronize()
pkg.lib().t2gpu_ldpc_profile(dec._h, None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bits, trials = dec.execute_dev(x); e1.record(); torch.cuda.synchronize()
out = (ctypes.c_longlong * 8)()
pkg.lib().t2gpu_ldpc_profile(dec._h, out)
names = ["parity check", "rendezvous", "PLAIN layers", "PAIR layers / slot3", "GENERIC / slot4", "slot5"]
tot = sum(out[:6])
print("launch %.3f ms, frames %d, avg updates %.2f" % (e0.elapsed_time(e1), frames, float((25 - trials.float()).mean())))
for n, v in zip(names, out):
    print("%-16s %14d cycles  %5.1f %%" % (n, v, 100.0 * v / max(tot, 1)))
print("workgroup lifetime sum %.3f ms over span %.3f ms -> average %.1f workgroups alive" % (out[6] / 1e5, out[7] / 1e5, out[6] / max(out[7], 1)))
lay = (ctypes.c_longlong * 64)()
if pkg.lib().t2gpu_ldpc_profile_layers(dec._h, lay) == 0 and sum(lay):
    tl = sum(lay)
    print("per-layer cycles of workgroup 0 (share of its layer time):")
    print(" ".join("%d:%.1f%%" % (i, 100.0 * lay[i] / tl) for i in range(64) if lay[i]))
    print("cycles per layer per sweep: " + " ".join("%d:%d" % (i, lay[i] // max(1, int((25 - trials.float()).max()))) for i in range(64) if lay[i]))
