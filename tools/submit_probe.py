import ctypes, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import sdr_receiver_dvb_t2_amd as pkg
l = pkg.lib()
l.t2gpu_ldpc_submit.restype = ctypes.c_int
l.t2gpu_ldpc_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
l.t2gpu_ldpc_collect.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
l.t2gpu_ldpc_create.restype = ctypes.c_void_p
hs = [l.t2gpu_ldpc_create(1, 3, 32, 0) for _ in range(8)]
llr = np.random.default_rng(1).integers(-20, 21, size=(32, 64800), dtype=np.int8)
import os
for cnt in (1, 2, 4, 8):
  for rnd in range(2):
    ts = []
    t00 = time.perf_counter()
    for h in hs[:cnt]:
        t0 = time.perf_counter(); rc = l.t2gpu_ldpc_submit(h, llr.ctypes.data, llr.size); ts.append((time.perf_counter() - t0) * 1e6)
        assert rc == 0, l.t2gpu_last_error()
    o = ctypes.c_void_p(); tr = ctypes.c_void_p(); n = ctypes.c_int()
    for h in hs[:cnt]:
        assert l.t2gpu_ldpc_collect(h, 1, ctypes.byref(o), ctypes.byref(tr), ctypes.byref(n)) == 0
    print("handles", cnt, "round", rnd, "submit us:", [round(x) for x in ts], "total ms %.2f" % ((time.perf_counter() - t00) * 1e3))
