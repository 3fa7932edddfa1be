#!/usr/bin/env python3
"""A chain of tiny launches on one stream (timed by events around every launch) while ANOTHER THREAD keeps one-batch LDPC decodes in
flight through the host-buffer slot (t2gpu_ldpc_submit / _collect on 7 handles: memsets, launch, results-to-host kernel, events) --
the slot-shaped path's situation while the frame before is being decoded. MODE=0: chain alone; 1: decodes from the other thread."""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import sdr_receiver_dvb_t2_amd as pkg
l = pkg.lib()
l.t2gpu_ldpc_submit.restype = ctypes.c_int
l.t2gpu_ldpc_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
l.t2gpu_ldpc_collect.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
l.t2gpu_ldpc_create.restype = ctypes.c_void_p
hs = [l.t2gpu_ldpc_create(1, 3, 32, 0) for _ in range(7)]
llr = np.random.default_rng(1).integers(-20, 21, size=(32, 64800), dtype=np.int8)
stop = False


def feeder():
    o = ctypes.c_void_p(); tr = ctypes.c_void_p(); n = ctypes.c_int()
    while not stop:
        for h in hs:
            assert l.t2gpu_ldpc_submit(h, llr.ctypes.data, llr.size) == 0, l.t2gpu_last_error()
        for h in hs:
            assert l.t2gpu_ldpc_collect(h, 1, ctypes.byref(o), ctypes.byref(tr), ctypes.byref(n)) == 0


chain = torch.cuda.Stream(priority=-1) if os.environ.get("CHAIN_STREAM", "1") == "1" else torch.cuda.default_stream()
x = torch.zeros(1024, device="cuda")
pinned = torch.zeros(1024).pin_memory()


ctx = pkg.t2_ofdm(5, 1, 6, 4, 0, 59, max_symbols=2)
spec_in = torch.randn(1, 32768, 2, device="cuda")
idx1 = torch.tensor([3], dtype=torch.int32, device="cuda")
KIND = os.environ.get("KIND", "add")


def one_launch():
    if KIND == "fft":
        ctx.fft_dev(spec_in)
    elif KIND == "eq":
        ctx.eq_data_dev(spec_in, idx1, want_sync=False)
    elif KIND == "sync":
        ctx.sym_sync_dev(0, 3, spec_in[0])
    else:
        x.add_(1.0)


def run_chain(n=200, host_store=False):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    with torch.cuda.stream(chain):
        for a, b in ev:
            a.record(chain)
            one_launch()
            if host_store:
                pinned.copy_(x, non_blocking=True)
            b.record(chain)
    chain.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / n
    d = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return "median %.1f us  p90 %.1f  max %.1f  mean %.1f; wall per launch %.1f us" % (d[len(d) // 2], d[9 * len(d) // 10], d[-1], sum(d) / len(d), wall)


torch.cuda.synchronize()
run_chain()
print("chain alone:                      ", run_chain())
print("chain + D2H of 4 KB alone:        ", run_chain(host_store=True))
th = threading.Thread(target=feeder); th.start()
time.sleep(0.05)
for _ in range(3):
    print("chain beside submits:             ", run_chain())
print("chain + D2H of 4 KB beside submits:", run_chain(host_store=True))
stop = True; th.join()
