"""Host-side mirrors of the reference's OFDM-side stage objects for the supported FFT sizes (16K / 32K, SISO).

``fast_fourier_transform`` /root/reference/src/DSP/fast_fourier_transform.h:27-72  (init :54-60, execute :62-70)
``data_symbol``            /root/reference/src/DVB_T2/data_symbol.h:25-66         (init :30, execute :31-32)

Both share one device context (tables, twiddles): ``t2_ofdm``. The reference's objects take their mode from a
``dvbt2_parameters`` struct filled while acquiring; here the same fields are constructor arguments."""
import ctypes

import numpy as np

from ._lib import lib, check, T2GpuError

FFTSIZE_16K, FFTSIZE_32K = 4, 5          # dvbt2_fft_mode_t (dvbt2_definition.h:116-126)


class t2_ofdm(object):
    INFO = ("fft_size", "k_total", "k_ext", "k_offset", "l_nulls", "c_p2", "c_data", "n_fc", "c_fc", "l_fc", "len_frame",
            "guard_interval_size")

    def __init__(self, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data, max_symbols=64, device=0):
        self._l = lib()
        self.args = (fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)
        info = (ctypes.c_int * 12)()
        check(self._l.t2gpu_ofdm_mode_info(*self.args, info), "t2gpu_ofdm_mode_info")
        for k, v in zip(self.INFO, info):
            setattr(self, k, v)
        self.n_p2 = 1
        self._h = self._l.t2gpu_ofdm_create(*self.args, max_symbols, device)
        if not self._h:
            raise T2GpuError("t2gpu_ofdm_create: " + self._l.t2gpu_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._l.t2gpu_ofdm_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- fast_fourier_transform::execute, batched
    def fft_dev(self, x):
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-2:] == (self.fft_size, 2)
        y = torch.empty_like(x)
        check(self._l.t2gpu_fft_execute_dev(self._h, x.data_ptr(), y.data_ptr(), x.numel() // (2 * self.fft_size),
                                            torch.cuda.current_stream(x.device).cuda_stream), "t2gpu_fft_execute_dev")
        return y

    def fft_stream_dev(self, stream_cells, first, frame_stride, per_frame, sym_stride, n_symbols, out=None):
        """FFT of n_symbols symbols read in place from a complex64 device stream (guard removal by addressing)."""
        import torch
        assert stream_cells.is_cuda and stream_cells.dtype == torch.complex64 and stream_cells.is_contiguous()
        last = first + ((n_symbols - 1) // per_frame) * frame_stride + ((n_symbols - 1) % per_frame) * sym_stride + self.fft_size
        assert last <= stream_cells.numel()
        y = out if out is not None else torch.empty((n_symbols, self.fft_size, 2), dtype=torch.float32, device=stream_cells.device)
        check(self._l.t2gpu_fft_execute_strided_dev(self._h, stream_cells.data_ptr(), first, frame_stride, per_frame, sym_stride,
                                                    y.data_ptr(), n_symbols, torch.cuda.current_stream(y.device).cuda_stream),
              "t2gpu_fft_execute_strided_dev")
        return y

    def fft(self, x):
        x = np.ascontiguousarray(x, np.complex64).reshape(-1, self.fft_size)
        y = np.empty_like(x)
        check(self._l.t2gpu_fft_execute(self._h, x.ctypes.data, y.ctypes.data, x.shape[0]), "t2gpu_fft_execute")
        return y

    # ---- data_symbol::execute, batched
    def eq_data_dev(self, symbols, symbol_index, want_sync=True, out=None):
        import torch
        n = symbols.shape[0]
        assert symbols.is_cuda and symbols.dtype == torch.float32 and symbols.is_contiguous()
        assert symbol_index.dtype == torch.int32 and symbol_index.numel() == n
        cells = out if out is not None else torch.empty((n, self.c_data, 2), dtype=torch.float32, device=symbols.device)
        sync = torch.empty((n, 2), dtype=torch.float32, device=symbols.device) if want_sync else None
        rc = self._l.t2gpu_eq_data_execute_dev(self._h, symbols.data_ptr(), symbol_index.data_ptr(), n, cells.data_ptr(),
                                               sync.data_ptr() if want_sync else None,
                                               torch.cuda.current_stream(symbols.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_eq_data_execute_dev")
        return cells, sync

    def eq_data_frames_dev(self, spec, n_frames, syms_per_frame, first_symbol, n_data_symbols, cells, cells_offset, want_sync=False):
        """Data symbols of whole frames in place: spec float32 [n_frames * syms_per_frame][fft_size][2] -> cells float32
        [n_frames][cells per frame][2] at cell offset cells_offset of every frame."""
        import torch
        assert spec.is_contiguous() and cells.is_contiguous() and cells.dim() == 3
        sync = torch.empty((n_frames * n_data_symbols, 2), dtype=torch.float32, device=spec.device) if want_sync else None
        rc = self._l.t2gpu_eq_data_frames_dev(self._h, spec.data_ptr(), n_frames, syms_per_frame, first_symbol, n_data_symbols,
                                              cells.data_ptr(), cells.shape[1], cells_offset, sync.data_ptr() if want_sync else None,
                                              torch.cuda.current_stream(spec.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_eq_data_frames_dev")
        return sync

    # ---- equaliser part of p2_symbol::execute, batched over frames
    def eq_p2_dev(self, symbols, want_sync=True):
        import torch
        n = symbols.shape[0]
        assert symbols.is_cuda and symbols.dtype == torch.float32 and symbols.is_contiguous()
        cells = torch.empty((n, self.c_p2, 2), dtype=torch.float32, device=symbols.device)
        sync = torch.empty((n, 2), dtype=torch.float32, device=symbols.device) if want_sync else None
        rc = self._l.t2gpu_eq_p2_execute_dev(self._h, symbols.data_ptr(), n, cells.data_ptr(),
                                             sync.data_ptr() if want_sync else None,
                                             torch.cuda.current_stream(symbols.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_eq_p2_execute_dev")
        return cells, sync

    # ---- fc_symbol::execute, batched over frames
    def eq_fc_dev(self, symbols, want_sync=True):
        import torch
        n = symbols.shape[0]
        assert symbols.is_cuda and symbols.dtype == torch.float32 and symbols.is_contiguous()
        cells = torch.empty((n, self.n_fc, 2), dtype=torch.float32, device=symbols.device)
        sync = torch.empty((n, 2), dtype=torch.float32, device=symbols.device) if want_sync else None
        rc = self._l.t2gpu_eq_fc_execute_dev(self._h, symbols.data_ptr(), n, cells.data_ptr(),
                                             sync.data_ptr() if want_sync else None,
                                             torch.cuda.current_stream(symbols.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_eq_fc_execute_dev")
        return cells, sync

    # ---- the by-reference outputs of {data,p2,fc}_symbol::execute from the pilots alone + the guard correlation, one symbol
    def sym_sync_dev(self, kind, idx_symbol, spectrum, buffered=None, guard=0, host=None, loop=None):
        """spectrum: float32 device tensor [fft_size][2]; buffered (optional): float32 device tensor [guard + fft_size][2] (guard
        first). Returns device tensors (cp4[4], sync[2]). host = (h_small, h_flag, seq): page-locked float32[8] / int32[1] tensors the
        kernel itself stores into, the word last."""
        import torch
        assert spectrum.is_cuda and spectrum.dtype == torch.float32 and spectrum.is_contiguous()
        cp4 = torch.zeros(4, dtype=torch.float32, device=spectrum.device)
        sync = torch.zeros(2, dtype=torch.float32, device=spectrum.device)
        hs, hf, seq = (host[0].data_ptr(), host[1].data_ptr(), host[2]) if host else (None, None, 0)
        rc = self._l.t2gpu_sym_sync_dev(self._h, kind, idx_symbol, spectrum.data_ptr(), buffered.data_ptr() if buffered is not None else None,
                                        guard, cp4.data_ptr(), sync.data_ptr(), hs, hf, seq, loop,
                                        torch.cuda.current_stream(spectrum.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_sym_sync_dev")
        return cp4, sync

    def fft_sym_sync_dev(self, kind, idx_symbol, buffered, guard, with_cp=True, tables=None, host=None, loop=None):
        """One buffered symbol (float32 device tensor [guard + fft_size][2], guard first): FFT of its useful part and, inside the FFT's
        last launch, sym_sync_dev's outputs. Returns (spectrum [fft_size][2], cp4[4], sync[2]). tables: the t2_ofdm whose pilot tables
        apply (default: this one)."""
        import torch
        assert buffered.is_cuda and buffered.dtype == torch.float32 and buffered.is_contiguous()
        spec = torch.empty((self.fft_size, 2), dtype=torch.float32, device=buffered.device)
        cp4 = torch.zeros(4, dtype=torch.float32, device=buffered.device)
        sync = torch.zeros(2, dtype=torch.float32, device=buffered.device)
        hs, hf, seq = (host[0].data_ptr(), host[1].data_ptr(), host[2]) if host else (None, None, 0)
        rc = self._l.t2gpu_fft_sym_sync_dev(self._h, (tables or self)._h, kind, idx_symbol, buffered.data_ptr(), guard, 1 if with_cp else 0,
                                            spec.data_ptr(), cp4.data_ptr(), sync.data_ptr(), hs, hf, seq, loop,
                                            torch.cuda.current_stream(buffered.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_fft_sym_sync_dev")
        return spec, cp4, sync

    def eq_data(self, idx_symbol, ofdm_cell):
        """Reference call shape: returns (cells complex64[c_data], sample_rate_offset, phase_offset)."""
        x = np.ascontiguousarray(ofdm_cell, np.complex64).reshape(self.fft_size)
        out = np.empty(self.c_data, np.complex64)
        sro, pho = ctypes.c_float(), ctypes.c_float()
        rc = self._l.t2gpu_eq_data_execute(self._h, idx_symbol, x.ctypes.data, out.ctypes.data, ctypes.byref(sro), ctypes.byref(pho))
        if rc < 0:
            check(rc, "t2gpu_eq_data_execute")
        return out, sro.value, pho.value
