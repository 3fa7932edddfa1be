"""Host-side mirror of the reference's ``ldpc_decoder`` stage object.

Reference: ``/root/reference/src/DVB_T2/ldpc_decoder.h:76-137`` (class), ``ldpc_decoder.cpp:157-301`` (``execute``).
The Qt slot ``execute(int* idx_plp_simd, l1_postsignalling, int len_in, int8_t* in)`` takes its code parameters from
``l1_post.plp[plp_id]`` (``plp_fec_type``, ``plp_cod``); here they are constructor arguments, the rest of the call
keeps the reference's meaning: ``in`` is ``[n_frames][fec_size]`` int8 LLRs (``len_in = fec_size * n_frames``), the
result is ``[n_frames][k_ldpc]`` hard bits, one per byte -- exactly what the reference emits through ``bit_bch``.
A batch that does not converge is *dropped* by the reference (``ldpc_decoder.cpp:264-268``); ``execute`` mirrors
that by returning ``None`` for such a batch.
"""
import ctypes

import numpy as np

from ._lib import lib, check, T2GpuError

# dvbt2_fectype_t / dvbt2_code_rate_t (dvbt2_definition.h:60-67,85-88)
FECFRAME_SHORT, FEC_FRAME_NORMAL = 0, 1
C1_2, C3_5, C2_3, C3_4, C4_5, C5_6 = range(6)

SIZEOF_SIMD = 32   # ldpc_decoder.h:28-32 (AVX2 build)
TRIALS = 25        # ldpc_decoder.h:63


class ldpc_decoder(object):
    def __init__(self, fec_type, code_rate, max_frames=SIZEOF_SIMD, device=0, group=SIZEOF_SIMD, trials=TRIALS):
        self._l = lib()
        self._h = self._l.t2gpu_ldpc_create(fec_type, code_rate, max_frames, device)
        if not self._h:
            raise T2GpuError("t2gpu_ldpc_create: " + self._l.t2gpu_last_error().decode())
        check(self._l.t2gpu_ldpc_configure(self._h, group, trials), "t2gpu_ldpc_configure")
        n, k, q, kb = (ctypes.c_int() for _ in range(4))
        check(self._l.t2gpu_ldpc_info(self._h, ctypes.byref(n), ctypes.byref(k), ctypes.byref(q), ctypes.byref(kb)),
              "t2gpu_ldpc_info")
        self.fec_size, self.k_ldpc, self.q_ldpc, self.k_bch = n.value, k.value, q.value, kb.value
        self.group, self.trials, self.max_frames, self.device = group, trials, max_frames, device

    def close(self):
        if getattr(self, "_h", None):
            self._l.t2gpu_ldpc_destroy(self._h)
            self._h = None

    __del__ = close

    def n_batches(self, n_frames):
        return (n_frames + self.group - 1) // self.group

    # ---- device-resident form: torch CUDA tensors in, torch CUDA tensors out, enqueued on the current stream
    def execute_dev(self, llr, want_llr=False):
        import torch
        assert llr.is_cuda and llr.dtype == torch.int8 and llr.is_contiguous() and llr.shape[-1] == self.fec_size
        n_frames = llr.numel() // self.fec_size
        bits = torch.empty((n_frames, self.k_ldpc), dtype=torch.uint8, device=llr.device)
        trials = torch.empty((self.n_batches(n_frames),), dtype=torch.int32, device=llr.device)
        llr_out = torch.empty_like(llr) if want_llr else None
        stream = torch.cuda.current_stream(llr.device).cuda_stream
        check(self._l.t2gpu_ldpc_execute_dev(self._h, llr.data_ptr(), n_frames, bits.data_ptr(),
                                              llr_out.data_ptr() if want_llr else None, trials.data_ptr(), stream),
              "t2gpu_ldpc_execute_dev")
        return (bits, trials, llr_out) if want_llr else (bits, trials)

    def wait_resident(self, stream=None):
        """Hold `stream` (default: torch's current one) until all workgroups of the decodes enqueued so far have started."""
        import torch
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        check(self._l.t2gpu_ldpc_wait_resident(self._h, s), "t2gpu_ldpc_wait_resident")

    def status(self):
        return self._l.t2gpu_ldpc_status(self._h)

    # ---- reference-shaped form: host buffers (numpy), synchronous
    def execute_host(self, llr):
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        n_frames = llr.size // self.fec_size
        bits = np.empty((n_frames, self.k_ldpc), dtype=np.uint8)
        trials = np.empty((self.n_batches(n_frames),), dtype=np.int32)
        check(self._l.t2gpu_ldpc_execute(self._h, llr.ctypes.data, llr.size, bits.ctypes.data, trials.ctypes.data),
              "t2gpu_ldpc_execute")
        return bits, trials

    def execute(self, idx_plp_simd, len_in, _in):
        """Reference call shape: returns the list of per-batch outputs the reference would emit through ``bit_bch``
        (``None`` where the reference drops the batch)."""
        llr = np.asarray(_in, dtype=np.int8).reshape(-1)[:len_in]
        bits, trials = self.execute_host(llr)
        out = []
        for b in range(len(trials)):
            out.append(bits[b * self.group:(b + 1) * self.group] if trials[b] >= 0 else None)
        return out
