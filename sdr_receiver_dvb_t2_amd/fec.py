"""Host-side mirrors of the reference's FEC-side stage objects around the LDPC decoder (the C ABI does the work).

``llr_demapper``       /root/reference/src/DVB_T2/llr_demapper.h:28-124      (execute :44-45, llr_demapper.cpp:132-158)
``time_deinterleaver`` /root/reference/src/DVB_T2/time_deinterleaver.h:27-102 (start / l1_dyn_execute / execute)
``bch_decoder``        /root/reference/src/DVB_T2/bch_decoder.h:26-59         (execute :41, bch_decoder.cpp:63-164)

The Qt slots take their mode from ``l1_post.plp[plp_id]``; here the same four integers (plp_mod, plp_fec_type, plp_cod,
plp_rotation) are constructor arguments. Buffers are torch CUDA tensors for the ``*_dev`` forms (no copies, current
stream) and numpy arrays for the reference-shaped host forms.
"""
import ctypes

import numpy as np

from ._lib import lib, check, T2GpuError

MOD_QPSK, MOD_16QAM, MOD_64QAM, MOD_256QAM = range(4)   # dvbt2_constellation_t (dvbt2_definition.h:69-74)


class llr_demapper(object):
    def __init__(self, plp_mod, plp_fec_type, plp_cod, plp_rotation, max_cells, device=0, saturate=False):
        self._l = lib()
        self._h = self._l.t2gpu_demap_create(plp_mod, plp_fec_type, plp_cod, plp_rotation, max_cells, device)
        if not self._h:
            raise T2GpuError("t2gpu_demap_create: " + self._l.t2gpu_last_error().decode())
        if saturate:        # extension: clamp instead of the reference's wrapping int8 cast
            check(self._l.t2gpu_demap_configure(self._h, 1), "t2gpu_demap_configure")
        self.fec_size = 64800 if plp_fec_type == 1 else 16200
        self.bits_per_cell = 2 * (plp_mod + 1)
        self.cells_per_fec = self.fec_size // self.bits_per_cell

    def close(self):
        if getattr(self, "_h", None):
            self._l.t2gpu_demap_destroy(self._h)
            self._h = None

    __del__ = close

    def execute_dev(self, cells, precision_override=0.0, out=None):
        """cells: CUDA float32 [n_cells, 2] (re, im). Returns (llr int8 [n_frames, fec_size], sums float32[3]); `out`: write the
        LLR frames there instead of a fresh tensor."""
        import torch
        assert cells.is_cuda and cells.dtype == torch.float32 and cells.is_contiguous()
        n_cells = cells.numel() // 2
        n_frames = n_cells // self.cells_per_fec
        if out is not None:
            assert out.is_contiguous() and out.dtype == torch.int8 and out.numel() == n_frames * self.fec_size
        llr = out if out is not None else torch.empty((n_frames, self.fec_size), dtype=torch.int8, device=cells.device)
        sums = torch.empty((3,), dtype=torch.float32, device=cells.device)
        stream = torch.cuda.current_stream(cells.device).cuda_stream
        rc = self._l.t2gpu_demap_execute_dev(self._h, cells.data_ptr(), n_cells, float(precision_override), llr.data_ptr(),
                                             sums.data_ptr(), stream)
        if rc < 0:
            check(rc, "t2gpu_demap_execute_dev")
        return llr, sums

    def stats_dev(self, cells, sums, precision_override=0.0):
        """First pass only: sum_s, sum_e and the LLR scale of this TI block into the float32[3] device tensor `sums`."""
        import torch
        rc = self._l.t2gpu_demap_stats_dev(self._h, cells.data_ptr(), cells.numel() // 2, float(precision_override), sums.data_ptr(),
                                           torch.cuda.current_stream(cells.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_demap_stats_dev")

    def stats_batch_dev(self, cells, sums, precision_override=0.0):
        """First pass for n TI blocks in one launch: cells float32 [n][block cells][2] (row stride free), sums float32 [n][>= 3]."""
        import torch
        assert cells.dim() == 3 and cells.stride(2) == 1 and cells.stride(1) == 2 and cells.stride(0) % 2 == 0
        assert sums.dim() == 2 and sums.shape[0] >= cells.shape[0] and sums.stride(1) == 1
        rc = self._l.t2gpu_demap_stats_batch_dev(self._h, cells.data_ptr(), cells.stride(0) // 2, cells.shape[0], cells.shape[1],
                                                 float(precision_override), sums.data_ptr(), sums.stride(0),
                                                 torch.cuda.current_stream(cells.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_demap_stats_batch_dev")

    def llr_dev(self, cells, sums, out):
        """Second pass only: LLR frames into `out` with the statistics `sums` of stats_dev."""
        import torch
        rc = self._l.t2gpu_demap_llr_dev(self._h, cells.data_ptr(), cells.numel() // 2, sums.data_ptr(), out.data_ptr(),
                                         torch.cuda.current_stream(cells.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_demap_llr_dev")

    def llr_batch_dev(self, cells, n_blocks, cells_per_block, sums, out):
        """LLR pass of n_blocks TI blocks (contiguous in `cells`, statistics rows in `sums` [n_blocks][>=3]) in one launch."""
        import torch
        rc = self._l.t2gpu_demap_llr_batch_dev(self._h, cells.data_ptr(), n_blocks, cells_per_block, sums.data_ptr(), sums.stride(0),
                                               out.data_ptr(), torch.cuda.current_stream(cells.device).cuda_stream)
        if rc < 0:
            check(rc, "t2gpu_demap_llr_batch_dev")
        return rc

    def execute(self, ti_block_size, time_deint_cell):
        """Reference call shape on host buffers; returns (llr [n_frames][fec_size], (sum_s, sum_e, precision))."""
        cells = np.ascontiguousarray(time_deint_cell, dtype=np.complex64).reshape(-1)[:ti_block_size]
        n_frames = ti_block_size // self.cells_per_fec
        llr = np.empty((n_frames, self.fec_size), dtype=np.int8)
        sums = np.empty(3, dtype=np.float32)
        rc = self._l.t2gpu_demap_execute(self._h, cells.ctypes.data, ti_block_size, llr.ctypes.data, sums.ctypes.data)
        if rc < 0:
            check(rc, "t2gpu_demap_execute")
        return llr, sums


class time_deinterleaver(object):
    def __init__(self, plp_mod, plp_fec_type, plp_num_blocks_max, device=0):
        self._l = lib()
        self._h = self._l.t2gpu_ti_create(plp_mod, plp_fec_type, plp_num_blocks_max, device)
        if not self._h:
            raise T2GpuError("t2gpu_ti_create: " + self._l.t2gpu_last_error().decode())
        self.cells_per_fec = self._l.t2gpu_ti_cells_per_fec(self._h)
        self.num_blocks = 0

    def close(self):
        if getattr(self, "_h", None):
            self._l.t2gpu_ti_destroy(self._h)
            self._h = None

    __del__ = close

    def l1_dyn(self, plp_num_blocks):
        """Geometry step of l1_dyn_execute (time_deinterleaver.cpp:268-286) for this T2 frame."""
        check(self._l.t2gpu_ti_begin(self._h, plp_num_blocks), "t2gpu_ti_begin")
        self.num_blocks = plp_num_blocks
        return plp_num_blocks * self.cells_per_fec

    def execute_dev(self, cells, out):
        import torch
        assert cells.is_cuda and out.is_cuda and cells.dtype == torch.float32 and out.dtype == torch.float32
        stream = torch.cuda.current_stream(cells.device).cuda_stream
        rc = self._l.t2gpu_ti_push_dev(self._h, cells.data_ptr(), cells.numel() // 2, out.data_ptr(), stream)
        if rc < 0:
            check(rc, "t2gpu_ti_push_dev")
        return rc == 1

    def execute_blocks_dev(self, cells, out):
        """cells, out: CUDA float32 [n][>= ti_block_size][2] views (row stride free): the same TI block of n frames in one launch."""
        import torch
        assert cells.is_cuda and out.is_cuda and cells.dtype == torch.float32 and out.dtype == torch.float32
        assert cells.dim() == 3 and out.dim() == 3 and cells.shape[0] == out.shape[0]
        assert cells.stride(2) == 1 and cells.stride(1) == 2 and out.stride(2) == 1 and out.stride(1) == 2
        assert cells.stride(0) % 2 == 0 and out.stride(0) % 2 == 0
        assert cells.shape[1] >= self.num_blocks * self.cells_per_fec and out.shape[1] >= self.num_blocks * self.cells_per_fec
        stream = torch.cuda.current_stream(cells.device).cuda_stream
        rc = self._l.t2gpu_ti_execute_blocks_dev(self._h, cells.data_ptr(), cells.stride(0) // 2, out.data_ptr(), out.stride(0) // 2,
                                                 cells.shape[0], stream)
        if rc < 0:
            check(rc, "t2gpu_ti_execute_blocks_dev")
        return rc

    def execute_blocks_stats_dev(self, cells, out, demapper, sums, precision_override=0.0):
        """execute_blocks_dev with the demapper's first pass folded in: sums float32 [n][>= 3] receives (sum_s, sum_e, precision) of
        every TI block, formed while the cells leave the de-interleaver (one pass over the cells instead of two)."""
        import torch
        assert cells.dim() == 3 and out.dim() == 3 and cells.shape[0] == out.shape[0] and cells.stride(1) == 2 and out.stride(1) == 2
        assert sums.dim() == 2 and sums.shape[0] >= cells.shape[0] and sums.stride(1) == 1 and sums.shape[1] >= 3
        stream = torch.cuda.current_stream(cells.device).cuda_stream
        rc = self._l.t2gpu_ti_execute_blocks_stats_dev(self._h, demapper._h, cells.data_ptr(), cells.stride(0) // 2, out.data_ptr(),
                                                       out.stride(0) // 2, cells.shape[0], float(precision_override), sums.data_ptr(),
                                                       sums.stride(0), stream)
        if rc < 0:
            check(rc, "t2gpu_ti_execute_blocks_stats_dev")
        return rc

    def execute(self, cells, out):
        cells = np.ascontiguousarray(cells, dtype=np.complex64).reshape(-1)
        assert out.dtype == np.complex64 and out.flags.c_contiguous
        rc = self._l.t2gpu_ti_push(self._h, cells.ctypes.data, cells.size, out.ctypes.data)
        if rc < 0:
            check(rc, "t2gpu_ti_push")
        return rc == 1


class ti_block(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("plp", "offset", "num_blocks", "size")]


def ti_frame_plan(plps, dyns, frame_cells, plp_state=0, max_blocks=1024):
    """TI blocks of one T2 frame's cell stream, in the order time_deinterleaver::execute emits ti_block for them
    (time_deinterleaver.cpp:296-312,337-371). plps / dyns: sequences of dicts with the L1-post fields the stage reads
    (plp_mod, plp_fec_type, plp_num_blocks_max, time_il_length, time_il_type / id, start, num_blocks) or the ctypes arrays
    of l1.l1_post_info. Returns ([(plp index, first cell, FEC blocks, cells)], PLP the frame ended in)."""
    from .l1 import dynamic_plp, l1_postsignalling_plp
    n = len(plps)
    P, D = (l1_postsignalling_plp * n)(), (dynamic_plp * n)()
    for i in range(n):
        for arr, src in ((P, plps[i]), (D, dyns[i])):
            if isinstance(src, dict):
                for k, v in src.items():
                    if hasattr(arr[i], k):
                        setattr(arr[i], k, int(v))
            else:
                ctypes.pointer(arr[i])[0] = src
    out = (ti_block * max_blocks)()
    st = ctypes.c_int(plp_state)
    rc = lib().t2gpu_ti_frame_plan(n, P, D, frame_cells, ctypes.byref(st), out, max_blocks)
    if rc < 0:
        check(rc, "t2gpu_ti_frame_plan")
    return [(out[k].plp, out[k].offset, out[k].num_blocks, out[k].size) for k in range(rc)], st.value


class bch_decoder(object):
    """The reference's BCH stage is a parity strip + BB descrambler (bch_decoder.cpp:136-142)."""

    def __init__(self, plp_fec_type, plp_cod):
        self._l = lib()
        self.fec_type, self.cod = plp_fec_type, plp_cod

    def execute_dev(self, bits):
        import torch
        assert bits.is_cuda and bits.dtype == torch.uint8 and bits.is_contiguous()
        n_frames = bits.shape[0]
        k_bch = {0: (7032, 9552, 10632, 11712, 12432, 13152), 1: (32208, 38688, 43040, 48408, 51648, 53840)}[self.fec_type][self.cod]
        out = torch.empty((n_frames, k_bch), dtype=torch.uint8, device=bits.device)
        stream = torch.cuda.current_stream(bits.device).cuda_stream
        rc = self._l.t2gpu_bch_descramble_dev(self.fec_type, self.cod, bits.data_ptr(), n_frames, out.data_ptr(), stream)
        if rc < 0:
            check(rc, "t2gpu_bch_descramble_dev")
        return out

    def execute_packed_dev(self, bits):
        """K-descramble-pack: [n][k_ldpc] device bits -> [n][k_bch / 8] device bytes (descrambled, MSB first): the byte assembly of
        bb_de_header.cpp:84-448 done on the device; t2gpu_bbdh_execute_packed reads the rows."""
        import torch
        assert bits.is_cuda and bits.dtype == torch.uint8 and bits.is_contiguous()
        n_frames = bits.shape[0]
        k_bch = {0: (7032, 9552, 10632, 11712, 12432, 13152), 1: (32208, 38688, 43040, 48408, 51648, 53840)}[self.fec_type][self.cod]
        out = torch.empty((n_frames, k_bch // 8), dtype=torch.uint8, device=bits.device)
        stream = torch.cuda.current_stream(bits.device).cuda_stream
        rc = self._l.t2gpu_bch_descramble_pack_dev(self.fec_type, self.cod, bits.data_ptr(), n_frames, out.data_ptr(), stream)
        if rc < 0:
            check(rc, "t2gpu_bch_descramble_pack_dev")
        return out

    def correct_dev(self, bits):
        """Apply the outer code in place on [n_frames][k_ldpc] device bits (opt-in: the reference does not, bch_decoder.cpp:136).
        Returns an int32 device tensor: bits corrected per frame, -1 = more than t errors (frame untouched)."""
        import torch
        assert bits.is_cuda and bits.dtype == torch.uint8 and bits.is_contiguous() and bits.dim() == 2
        status = torch.empty(bits.shape[0], dtype=torch.int32, device=bits.device)
        stream = torch.cuda.current_stream(bits.device).cuda_stream
        rc = self._l.t2gpu_bch_decode_dev(self.fec_type, self.cod, bits.data_ptr(), bits.shape[0], status.data_ptr(), stream)
        if rc < 0:
            check(rc, "t2gpu_bch_decode_dev")
        return status

    def correct(self, words):
        """Host form of correct_dev: returns (corrected copy of [n_frames][k_ldpc] bits, status)."""
        out = np.ascontiguousarray(words, dtype=np.uint8).copy()
        status = np.empty(out.shape[0], dtype=np.int32)
        rc = self._l.t2gpu_bch_decode(self.fec_type, self.cod, out.ctypes.data, out.shape[0], status.ctypes.data)
        if rc < 0:
            check(rc, "t2gpu_bch_decode")
        return out, status

    def execute(self, len_in, _in):
        bits = np.ascontiguousarray(_in, dtype=np.uint8).reshape(-1)[:len_in]
        k_ldpc = {0: (7200, 9720, 10800, 11880, 12600, 13320), 1: (32400, 38880, 43200, 48600, 51840, 54000)}[self.fec_type][self.cod]
        n_frames = len_in // k_ldpc
        k_bch = {0: (7032, 9552, 10632, 11712, 12432, 13152), 1: (32208, 38688, 43040, 48408, 51648, 53840)}[self.fec_type][self.cod]
        out = np.empty((n_frames, k_bch), dtype=np.uint8)
        rc = self._l.t2gpu_bch_descramble(self.fec_type, self.cod, bits.ctypes.data, n_frames, out.ctypes.data)
        if rc < 0:
            check(rc, "t2gpu_bch_descramble")
        return out
