#!/usr/bin/env python3
"""Generate tests/golden/dsp_golden.npz from the REFERENCE's own Qt-free DSP classes (oracle/_ref/libref_dsp_strict.so and
libref_dsp.so, i.e. /root/reference/src/DSP/{filter_decimator.h,interpolator_farrow.hh,loop_filters.hh,buffers.hh} compiled
unmodified by oracle/Makefile). Run in the build container:

    make -C oracle && python tests/golden/make_dsp_golden.py

Stored: a seeded complex input and what the reference classes made of it -- decimator output (call lengths that move the
decimation phase), Farrow output for four ratios, exponential-averager trace, both PI loop filters' outputs, the running sums /
delays of the P1 correlator. "strict" = built without fast-math (operations in source order), "fast" = the reference's -Ofast."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

rng = np.random.Generator(np.random.PCG64(20250614))
x = ((rng.standard_normal(6000) + 1j * rng.standard_normal(6000)) * 0.2).astype(np.complex64)
out = {"x": x}
LENS = [1, 2, 63, 64, 65, 1000, 4805]
out["decim_lens"] = np.array(LENS)
for tag, strict in (("strict", True), ("fast", False)):
    d = ol.OraDecim(ref=True, strict=strict)
    pos, parts = 0, []
    for n in LENS:
        parts.append(d(x[pos:pos + n]))
        pos += n
    out["decim_" + tag] = np.concatenate(parts)
    del d
    for k, r in enumerate((0.5, 0.5 - 24e-9, 0.546875, 0.4571)):
        f = ol.OraFarrow(ref=True, strict=strict)
        out["farrow%d_%s" % (k, tag)] = np.concatenate([f(x[:3000], r), f(x[3000:], r)])
out["farrow_ratios"] = np.array([0.5, 0.5 - 24e-9, 0.546875, 0.4571])
r = ol.ref_dsp()
xr = (x.real + 0.01).astype(np.float32)
r.ref_avg_new.restype = ctypes.c_void_p
r.ref_avg_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
h = r.ref_avg_new()
avg = np.zeros_like(xr)
r.ref_avg_run(h, len(xr), xr.ctypes.data, avg.ctypes.data)
out["avg_in"], out["avg_out"] = xr, avg
r.ref_pi_new.restype = ctypes.c_void_p
r.ref_pi_new.argtypes = [ctypes.c_int]
r.ref_pi_step.restype = ctypes.c_float
r.ref_pi_step.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
for which, lim in ((0, 6.2831855), (1, 1.0 / 32768)):
    hp = r.ref_pi_new(which)
    e = (rng.standard_normal(500) * lim * 0.3).astype(np.float32)
    out["pi%d_in" % which] = e
    out["pi%d_out" % which] = np.array([r.ref_pi_step(which, hp, float(v), lim) for v in e], np.float32)
    out["pi%d_lim" % which] = np.float32(lim)
rs = ol.ref_dsp(True)
for which, ln in ((0, 482), (1, 542)):
    o = np.zeros_like(x)
    rs.ref_sum_run(which, len(x), x.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p))
    out["sum%d" % ln] = o
np.savez_compressed(os.path.join(HERE, "dsp_golden.npz"), **out)
print("wrote dsp_golden.npz:", {k: v.shape for k, v in out.items()})
