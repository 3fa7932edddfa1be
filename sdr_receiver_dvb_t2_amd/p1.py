"""P1 preamble detector: host-side mirror of ``p1_symbol`` (src/DVB_T2/p1_symbol.h). All compute is in libt2gpu.so."""
import ctypes

import numpy as np

from ._lib import T2GpuError, lib


class p1_result(ctypes.Structure):
    _fields_ = [("detected", ctypes.c_int32), ("idx_buffer_sym", ctypes.c_int32), ("p1_decoded", ctypes.c_int32),
                ("preamble", ctypes.c_int32), ("fft_mode", ctypes.c_int32), ("s1", ctypes.c_int32), ("s2", ctypes.c_int32),
                ("shift", ctypes.c_int32), ("a_part_clipped", ctypes.c_int32), ("max_correlation", ctypes.c_float),
                ("arg_max", ctypes.c_float * 2), ("coarse_freq_offset", ctypes.c_double)]


class p1_symbol(object):
    def __init__(self, max_samples=1 << 20, device=0):
        self._l = lib()
        self.h = self._l.t2gpu_p1_create(max_samples, device)
        if not self.h:
            raise T2GpuError("t2gpu_p1_create: " + self._l.t2gpu_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self._l.t2gpu_p1_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        self._l.t2gpu_p1_reset(self.h)

    def set_serial_detector(self, on=True):
        """Tests: the detector's state machine sample by sample only (the closed-form stretches off)."""
        self._l.t2gpu_p1_set_serial_detector(self.h, int(on))

    def execute(self, x, consume=0, gain_changed=False, level_detect=0.0, reset=False):
        """x: complex64 host array. Returns (detected, consume, result) like the reference's execute()."""
        x = np.ascontiguousarray(x, np.complex64)
        c = ctypes.c_int(consume)
        res = p1_result()
        rc = self._l.t2gpu_p1_execute(self.h, int(gain_changed), float(level_detect), len(x), x.ctypes.data, ctypes.byref(c),
                                      int(reset), ctypes.byref(res))
        if rc < 0:
            raise T2GpuError("t2gpu_p1_execute: " + self._l.t2gpu_last_error().decode())
        return bool(rc), c.value, res

    def execute_dev(self, x, consume=0, gain_changed=False, level_detect=0.0, reset=False, stream=None):
        """x: complex64 torch device tensor."""
        import torch
        c = ctypes.c_int(consume)
        res = p1_result()
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = self._l.t2gpu_p1_execute_dev(self.h, int(gain_changed), float(level_detect), x.numel(), x.data_ptr(), ctypes.byref(c),
                                          int(reset), ctypes.byref(res), s)
        if rc < 0:
            raise T2GpuError("t2gpu_p1_execute_dev: " + self._l.t2gpu_last_error().decode())
        return bool(rc), c.value, res

    def execute_batch_dev(self, x, win_start, win_len, gain_changed=False, level_detect=0.0, reset=False, stream=None):
        """x: complex64 torch device tensor (the sample stream); windows searched independently in one launch sequence.
        Returns (list of p1_result, consumed per window)."""
        import torch
        ws = np.ascontiguousarray(win_start, np.int64)
        wl = np.ascontiguousarray(win_len, np.int32)
        assert len(ws) == len(wl) and int((ws + wl).max()) <= x.numel()
        res = (p1_result * len(ws))()
        cons = np.zeros(len(ws), np.int32)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = self._l.t2gpu_p1_execute_batch_dev(self.h, int(gain_changed), float(level_detect), x.data_ptr(), len(ws), ws.ctypes.data,
                                                wl.ctypes.data, int(reset), ctypes.byref(res), cons.ctypes.data, s)
        if rc < 0:
            raise T2GpuError("t2gpu_p1_execute_batch_dev: " + self._l.t2gpu_last_error().decode())
        return list(res), cons

    def debug(self, n):
        corr = np.zeros(n, np.float32)
        fft = np.zeros(1024, np.complex64)
        got = self._l.t2gpu_p1_debug(self.h, corr.ctypes.data, n, fft.ctypes.data)
        return corr[:max(got, 0)], fft
