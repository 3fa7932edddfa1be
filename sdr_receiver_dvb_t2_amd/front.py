"""Sample-rate front end: host-side mirror of the reference objects for this part of the path.

``front_end``  ~ the per-sample loop of ``dvbt2_demodulator::execute`` with its ``interpolator_farrow`` and
                 ``filter_decimator`` members (src/DVB_T2/dvbt2_demodulator.cpp:145-254)
``sync_loops`` ~ the tracking updates of ``symbol_acquisition`` (:328-330,429-439)
``cp_correlate_dev`` ~ the guard-interval correlation (:321-327)
All compute happens in libt2gpu.so on the GPU; there is no CPU path."""
import ctypes

import numpy as np

from ._lib import T2GpuError, lib

SAMPLE_RATE = float(np.float32(1.0) / (np.float32(1.0e-6) * np.float32(7.0) / np.float32(64.0)))   # dvbt2_definition.h:29-30


def _err(what):
    raise T2GpuError("%s: %s" % (what, lib().t2gpu_last_error().decode()))


class front_end(object):
    def __init__(self, id_device=0, sample_rate=SAMPLE_RATE, max_samples=1 << 21, device=0):
        self._l = lib()
        self.h = self._l.t2gpu_front_create(id_device, sample_rate, max_samples, device)
        if not self.h:
            _err("t2gpu_front_create")
        self.stride = 2 if id_device == 1 else 1
        self.max_samples = max_samples
        r, m = ctypes.c_double(), ctypes.c_double()
        self._l.t2gpu_front_resample(self.h, ctypes.byref(r), ctypes.byref(m))
        self.resample, self.max_resample = r.value, m.value

    def close(self):
        if getattr(self, "h", None):
            self._l.t2gpu_front_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        if self._l.t2gpu_front_reset(self.h) != 0:
            _err("t2gpu_front_reset")

    def set_iq(self, c1, c2):
        if self._l.t2gpu_front_set_iq(self.h, float(c1), float(c2)) != 0:
            _err("t2gpu_front_set_iq")

    @staticmethod
    def _loops(n_chunks, pe, fe, rs):
        pe = np.ascontiguousarray(np.zeros(n_chunks) if pe is None else pe, np.float32)
        fe = np.ascontiguousarray(np.zeros(n_chunks) if fe is None else fe, np.float32)
        rs = None if rs is None else np.ascontiguousarray(rs, np.float64)
        return pe, fe, rs

    def execute(self, i_in, q_in, chunk_len, phase_est_filtered=None, frequency_est_filtered=None, arbitrary_resample=None):
        """Host arrays in, host arrays out: (decimated complex64 stream, per-chunk output lengths)."""
        i_in = np.ascontiguousarray(i_in, np.int16)
        q_in = np.ascontiguousarray(q_in, np.int16)
        cl = np.ascontiguousarray(chunk_len, np.int32)
        pe, fe, rs = self._loops(len(cl), phase_est_filtered, frequency_est_filtered, arbitrary_resample)
        cap = int(cl.sum() * 1.01 / min(self.resample, 1.0) / 2) + 64
        out = np.zeros(cap, np.complex64)
        col = np.zeros(len(cl), np.int32)
        n = self._l.t2gpu_front_execute(self.h, len(cl), cl.ctypes.data, pe.ctypes.data, fe.ctypes.data,
                                        rs.ctypes.data if rs is not None else None, i_in.ctypes.data, q_in.ctypes.data,
                                        out.ctypes.data, cap, col.ctypes.data)
        if n < 0:
            _err("t2gpu_front_execute")
        return out[:n].copy(), col

    def execute_dev(self, d_i, d_q, chunk_len, out, phase_est_filtered=None, frequency_est_filtered=None,
                    arbitrary_resample=None, stream=None):
        """torch int16 device tensors in; writes into the complex64 device tensor ``out``; returns (cells, per-chunk)."""
        import torch
        cl = np.ascontiguousarray(chunk_len, np.int32)
        pe, fe, rs = self._loops(len(cl), phase_est_filtered, frequency_est_filtered, arbitrary_resample)
        col = np.zeros(len(cl), np.int32)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        n = self._l.t2gpu_front_execute_dev(self.h, len(cl), cl.ctypes.data, pe.ctypes.data, fe.ctypes.data,
                                            rs.ctypes.data if rs is not None else None, d_i.data_ptr(), d_q.data_ptr(),
                                            out.data_ptr(), out.numel(), col.ctypes.data, s)
        if n < 0:
            _err("t2gpu_front_execute_dev")
        return n, col

    # ---- the loop on the device (include/t2gpu.h)
    def loop_dev(self):
        return self._l.t2gpu_front_loop_dev(self.h)

    def loop_begin(self, state10, stream=None):
        import torch
        v = np.ascontiguousarray(state10, np.float32)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        if self._l.t2gpu_front_loop_begin(self.h, v.ctypes.data, s) != 0:
            _err("t2gpu_front_loop_begin")

    def execute_loop_dev(self, d_i, d_q, chunk, out, arbitrary_resample, stream=None):
        import torch
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        n = self._l.t2gpu_front_execute_loop_dev(self.h, int(chunk), float(arbitrary_resample), d_i.data_ptr(), d_q.data_ptr(), out.data_ptr(),
                                                 out.numel(), s)
        if n == -1:
            _err("t2gpu_front_execute_loop_dev")
        return n

    def loop_follow(self, pe, fe):
        if self._l.t2gpu_front_loop_follow(self.h, float(pe), float(fe)) != 0:
            _err("t2gpu_front_loop_follow")

    def loop_read(self, stream=None):
        import torch
        v = np.zeros(8, np.float32)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        if self._l.t2gpu_front_loop_read(self.h, v.ctypes.data, s) != 0:
            _err("t2gpu_front_loop_read")
        return dict(phase_nco=v[0], frequency_nco=v[1], pe=v[2], fe=v[3], frequency_est_filtered=v[4], f_int=v[5], p_int=v[6], error=int(v[7]))

    def state(self):
        v = np.zeros(8, np.float32)
        if self._l.t2gpu_front_state(self.h, v.ctypes.data) != 0:
            _err("t2gpu_front_state")
        return dict(dc_re=v[0], dc_im=v[1], c1=v[2], c2=v[3], phase_nco=v[4], frequency_nco=v[5], level_detect=v[6], x1=v[7])

    def debug_stream(self, which, cap):
        out = np.zeros(cap, np.complex64)
        n = self._l.t2gpu_front_debug_stream(self.h, which, out.ctypes.data, cap)
        if n < 0:
            _err("t2gpu_front_debug_stream")
        return out[:n].copy()

    # stand-alone stages, call shape of the reference classes
    def decimate(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.zeros(len(x) // 2 + 1, np.complex64)
        n = self._l.t2gpu_decim_execute(self.h, len(x), x.ctypes.data, out.ctypes.data)
        if n < 0:
            _err("t2gpu_decim_execute")
        return out[:n].copy()

    def farrow(self, x, arbitrary_resample):
        x = np.ascontiguousarray(x, np.complex64)
        cap = int(len(x) / max(arbitrary_resample, 0.01)) + 8
        out = np.zeros(cap, np.complex64)
        n = self._l.t2gpu_farrow_execute(self.h, len(x), x.ctypes.data, arbitrary_resample, out.ctypes.data, cap)
        if n < 0:
            _err("t2gpu_farrow_execute")
        return out[:n].copy()


def cp_correlate_dev(symbols, fft_size, guard, stream=None):
    """symbols: complex64 device tensor [n][guard + fft_size] (guard first). Returns float32 device tensor [n][4] =
    (sum.re, sum.im, frequency_est, 0)."""
    import torch
    n = symbols.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=symbols.device)
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    if lib().t2gpu_cp_correlate_dev(symbols.data_ptr(), n, fft_size, guard, out.data_ptr(), s) != 0:
        _err("t2gpu_cp_correlate_dev")
    return out


class sync_loops(object):
    def __init__(self, sample_rate=SAMPLE_RATE):
        self._l = lib()
        self.h = self._l.t2gpu_sync_create(sample_rate)

    def close(self):
        if getattr(self, "h", None):
            self._l.t2gpu_sync_destroy(self.h)
            self.h = None

    __del__ = close

    def frequency(self, frequency_est, fft_size):
        self._l.t2gpu_sync_frequency(self.h, frequency_est, fft_size)

    def symbol(self, phase_est, sample_rate_est):
        self._l.t2gpu_sync_symbol(self.h, phase_est, sample_rate_est)

    def export(self):
        """The filters' state as t2gpu_front_loop_begin takes it (float32[10]; [2] is the caller's tuner)."""
        v = np.zeros(10, np.float32)
        self._l.t2gpu_sync_export(self.h, v.ctypes.data)
        return v

    def get(self):
        v = np.zeros(4, np.float64)
        self._l.t2gpu_sync_get(self.h, v.ctypes.data)
        return dict(phase_est_filtered=np.float32(v[0]), frequency_est_filtered=np.float32(v[1]),
                    sample_rate_est_filtered=v[2], arbitrary_resample=v[3])


def plan_nco(frequency_nco, n, frequency_est_filtered):
    """Host only: (values[n], new accumulator, number of runs) -- exact expansion of the NCO run table."""
    acc = ctypes.c_float(frequency_nco)
    vals = np.zeros(n, np.float32)
    nr = ctypes.c_int()
    if lib().t2gpu_plan_nco(ctypes.byref(acc), n, frequency_est_filtered, vals.ctypes.data, ctypes.byref(nr)) != 0:
        _err("t2gpu_plan_nco")
    return vals, acc.value, nr.value


def plan_farrow(x1, n, arbitrary_resample):
    """Host only: (counts[n], positions[n], new x1, total outputs, number of runs)."""
    acc = ctypes.c_float(x1)
    cnt = np.zeros(n, np.int32)
    pos = np.zeros(n, np.float32)
    nr = ctypes.c_int()
    total = lib().t2gpu_plan_farrow(ctypes.byref(acc), n, arbitrary_resample, cnt.ctypes.data, pos.ctypes.data, ctypes.byref(nr))
    if total < 0:
        _err("t2gpu_plan_farrow")
    return cnt, pos, acc.value, total, nr.value
