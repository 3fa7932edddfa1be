"""demod -> TS chain over the C ABI: guard-removed OFDM symbols of whole T2 frames in, transport-stream bytes out.

Mirrors the call sequence of the reference's pipeline objects (TI type 0; one PLP, or several PLPs of one modulation / code):
``dvbt2_demodulator::symbol_acquisition`` (/root/reference/src/DVB_T2/dvbt2_demodulator.cpp:332-375: fft->execute, then
p2_demodulator / data_demodulator ->execute) -> ``time_deinterleaver::execute`` -> ``llr_demapper::execute`` ->
``ldpc_decoder::execute`` -> ``bch_decoder::execute`` -> ``bb_de_header::execute``. All stages but the last run on the GPU
on one stream with device-resident buffers (torch tensors are the allocator, nothing is computed by torch); BBFRAME
de-framing is host code in libt2gpu.so as it is host code in the reference. The mode (what L1-pre / L1-post signal) is
given by the caller; P1 / L1 acquisition are not part of this class."""
import ctypes

import numpy as np

from ._lib import lib, check
from .fec import bch_decoder, llr_demapper, time_deinterleaver, ti_frame_plan
from .ldpc import ldpc_decoder
from .ofdm import t2_ofdm

L1_PRE_CELL = 1840      # dvbt2_definition.h:58


def ts_from_bits(bits_host, trials_host, group=32, need_plp=0, tags=None, bbdh=None):
    """Descrambled BBFRAME bits [frames][k_bch] -> TS bytes through the library's de-framer (t2gpu_bbdh_*); SIMD batches the LDPC
    gave up on (trials -1) are dropped as the reference drops them (ldpc_decoder.cpp:264-268)."""
    l = lib()
    own = bbdh is None
    h = l.t2gpu_bbdh_create(need_plp) if own else bbdh
    out = []
    buf = np.zeros(bits_host.shape[1] // 8 + 400, np.uint8)
    err = ctypes.c_int(0)
    for i in range(bits_host.shape[0]):
        if trials_host[i // group] < 0:
            continue
        n = l.t2gpu_bbdh_execute(h, need_plp if tags is None else tags[i], bits_host.shape[1], bits_host[i].ctypes.data, buf.ctypes.data,
                                 buf.size, ctypes.byref(err))
        if n > 0:
            out.append(buf[:n].copy())
    if own:
        l.t2gpu_bbdh_destroy(h)
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


class t2_chain(object):
    def __init__(self, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data, l1_post_size,
                 plp_mod, plp_fec_type, plp_cod, plp_rotation, plp_num_blocks, max_frames=4, device=0, ldpc_group=32,
                 ldpc_trials=25, saturate_llr=False, time_il_length=1, plps=None, need_plp=0, outer_code=False):
        """plps (optional): several PLPs in the frame, a list of dicts with num_blocks and start (L1-post dynamic: PLP_NUM_BLOCKS,
        PLP_START) and optionally plp_rotation, time_il_length, plp_num_blocks_max, id; list position = PLP index. All PLPs
        share plp_mod / plp_fec_type / plp_cod: the reference decodes a SIMD batch with the code of its first frame whichever
        PLPs the other 31 belong to (ldpc_decoder.cpp:173-174), fills one batch per modulation (llr_demapper.cpp:172,242,378,549)
        and finds a PLP's end with one cell count (time_deinterleaver.cpp:273-274). need_plp: the PLP whose BBFRAMEs become the
        transport stream (bb_de_header::set_out, bb_de_header.cpp:500-525). Without plps: one PLP starting at cell 0 with
        plp_num_blocks FEC blocks per frame."""
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.max_frames = max_frames
        self.ofdm = t2_ofdm(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data,
                            max_symbols=max_frames * (n_data + 1), device=device)
        o = self.ofdm
        self.n_sym = o.n_p2 + n_data                            # P2, data symbols, frame-closing symbol when the mode has one
        self.n_dat = n_data - o.l_fc
        self.p2_skip = L1_PRE_CELL + l1_post_size               # time_deinterleaver.cpp:46,296-300
        self.frame_cells = (o.c_p2 - self.p2_skip) + self.n_dat * o.c_data + o.l_fc * o.n_fc
        self.fec_size = 64800 if plp_fec_type == 1 else 16200
        self.cells_per_fec = self.fec_size // (2 * (plp_mod + 1))
        # TI blocks of one T2 frame in the order the reference emits them (time_il_type 0: time_il_length blocks per PLP and
        # frame, the later blocks take the remainder, time_deinterleaver.cpp:275-285; PLP after PLP, :357-368)
        if plps is None:
            plps = [dict(num_blocks=plp_num_blocks, start=0, plp_rotation=plp_rotation, time_il_length=max(1, time_il_length))]
        P = [dict(plp_mod=plp_mod, plp_fec_type=plp_fec_type, plp_cod=plp_cod, plp_rotation=q.get("plp_rotation", plp_rotation),
                  time_il_length=q.get("time_il_length", 1), time_il_type=0,
                  plp_num_blocks_max=q.get("plp_num_blocks_max", q["num_blocks"])) for q in plps]
        D = [dict(id=q.get("id", i), start=q["start"], num_blocks=q["num_blocks"]) for i, q in enumerate(plps)]
        self.plan, _ = ti_frame_plan(P, D, self.frame_cells)
        assert self.plan, "no TI block fits the frame"
        self.ti_blocks = [b[2] for b in self.plan]
        self.frame_tags = [b[0] for b in self.plan for _ in range(b[2])]     # PLP of every FEC frame of a T2 frame
        plp_num_blocks = self.num_blocks = len(self.frame_tags)              # FEC frames per T2 frame, all PLPs
        self.need_plp = need_plp
        self.ti = [time_deinterleaver(plp_mod, plp_fec_type, q["plp_num_blocks_max"], device) for q in P]   # one per PLP
        by_rotation = {}
        for q in P:
            if q["plp_rotation"] not in by_rotation:
                by_rotation[q["plp_rotation"]] = llr_demapper(plp_mod, plp_fec_type, plp_cod, q["plp_rotation"],
                                                             max(b[3] for b in self.plan), device, saturate=saturate_llr)
        self.demaps = [by_rotation[q["plp_rotation"]] for q in P]
        self.demap = self.demaps[self.plan[0][0]]
        self.ldpc = ldpc_decoder(plp_fec_type, plp_cod, max_frames=max_frames * plp_num_blocks + 64, device=device,
                                 group=ldpc_group, trials=ldpc_trials)
        self.bch = bch_decoder(plp_fec_type, plp_cod)
        # opt-in BCH check / correction in front of the descrambler (the reference has none, bch_decoder.cpp:136); status of the
        # last call (bits corrected per FEC frame, -1 = beyond t) in self.outer_code_status
        self.outer_code, self.outer_code_status = bool(outer_code), None
        self.group = ldpc_group
        self._l = lib()
        self._bbdh = self._l.t2gpu_bbdh_create(need_plp)
        f32 = torch.float32
        self.cells = torch.zeros((max_frames, self.frame_cells, 2), dtype=f32, device=self.dev)
        self.p2_cells = torch.empty((max_frames, o.c_p2, 2), dtype=f32, device=self.dev)
        self.ti_out = torch.zeros((max_frames, plp_num_blocks * self.cells_per_fec, 2), dtype=f32, device=self.dev)
        self.llr = torch.empty((max_frames * plp_num_blocks + 64, self.fec_size), dtype=torch.int8, device=self.dev)
        self.carry = 0                                           # FEC frames waiting for a full SIMD batch
        self.tags, self.last_tags = [], []                       # PLP of the waiting frames / of the frames last handed out
        self.time_ldpc, self.ldpc_events = False, []             # bench: HIP events around the LDPC launch
        idx = np.tile(np.arange(1, 1 + self.n_dat, dtype=np.int32), max_frames)
        self.sym_index = torch.from_numpy(idx).to(self.dev)

    def close(self):
        if getattr(self, "_bbdh", None):
            self._l.t2gpu_bbdh_destroy(self._bbdh)
            self._bbdh = None

    def demod_dev(self, symbols, flush=False):
        """symbols: CUDA float32 [F][n_sym][fft_size][2], F <= max_frames. Returns (bits uint8 [frames][k_bch] on the
        device, trials-left int32 per SIMD batch) for the FEC frames whose batch of `group` completed (all of them with
        flush=True: the tail batch is then decoded short, which the reference never does)."""
        o = self.ofdm
        F = symbols.shape[0]
        assert F <= self.max_frames and symbols.shape[1] == self.n_sym
        spec = o.fft_dev(symbols.reshape(F * self.n_sym, o.fft_size, 2)).reshape(F, self.n_sym, o.fft_size, 2)
        return self.demod_spectrum_dev(spec, flush)

    def demod_stream_dev(self, stream, first, frame_stride, n_frames, flush=False):
        """The same from the decimated sample stream (complex64 device tensor): frame f's first useful P2 sample is at
        first + f * frame_stride, symbols follow every guard + fft_size samples; the guard interval is skipped by addressing
        (symbol_acquisition's memcpy from buffer_sym + guard_interval_size, dvbt2_demodulator.cpp:332-333)."""
        o = self.ofdm
        F = n_frames
        assert F <= self.max_frames
        spec = o.fft_stream_dev(stream, first, frame_stride, self.n_sym, o.fft_size + o.guard_interval_size, F * self.n_sym)
        return self.demod_spectrum_dev(spec.reshape(F, self.n_sym, o.fft_size, 2), flush)

    def demod_spectrum_dev(self, spec, flush=False):
        """spec: CUDA float32 [F][n_sym][fft_size][2], the fft-shifted spectra of whole frames."""
        F = self.spectrum_to_cells(spec)
        return self.demod_cells_dev(F, flush)

    def spectrum_to_cells(self, spec):
        """Equalisers + frequency de-interleavers: fills self.cells[:F] with the PLP cells of every frame; returns F."""
        o = self.ofdm
        F = spec.shape[0]
        # P2: equalise, drop the L1 cells, PLP cells go to the head of the frame's cell stream
        p2, _ = o.eq_p2_dev(spec[:, 0].contiguous(), want_sync=False)    # open loop: the feedback values are not consumed
        self.cells[:F, :o.c_p2 - self.p2_skip] = p2[:, self.p2_skip:]
        # data symbols: equalised cells land directly behind, symbol after symbol
        nd = self.n_dat
        a = o.c_p2 - self.p2_skip
        if spec.is_contiguous():            # straight from the frames' spectra into the frames' cell streams
            o.eq_data_frames_dev(spec, F, self.n_sym, 1, nd, self.cells, a)
        else:
            data = spec[:, 1:1 + nd].contiguous().reshape(F * nd, o.fft_size, 2)
            cells, _ = o.eq_data_dev(data, self.sym_index[:F * nd], want_sync=False)
            self.cells[:F, a:a + nd * o.c_data] = cells.reshape(F, nd * o.c_data, 2)
        if o.l_fc:                                               # frame-closing symbol: its n_fc cells end the frame's stream
            fc, _ = o.eq_fc_dev(spec[:, 1 + nd].contiguous(), want_sync=False)
            self.cells[:F, a + nd * o.c_data:] = fc
        return F

    def demod_cells_dev(self, F, flush=False):
        """From the equalised, frequency-de-interleaved cells of F frames already in self.cells[:F] (PLP cells of P2, then of every
        data symbol, then of the frame-closing symbol) to descrambled BBFRAME bits."""
        torch = self.torch
        c0 = 0
        for plp, off, nbk, n in self.plan:                       # TI block k of every frame in one launch
            self.ti[plp].l1_dyn(nbk)
            self.ti[plp].execute_blocks_dev(self.cells[:F, off:off + n], self.ti_out[:F, c0:c0 + n])
            c0 += n
        for f in range(F):
            a, c0 = self.carry + f * self.num_blocks, 0
            for plp, off, nbk, n in self.plan:                   # one TI block after the other, each with its own SNR estimate
                self.demaps[plp].execute_dev(self.ti_out[f, c0:c0 + n], out=self.llr[a:a + nbk])
                a, c0 = a + nbk, c0 + n
            self.tags += self.frame_tags
        total = self.carry + F * self.num_blocks
        ready = total if flush else (total // self.group) * self.group
        if ready == 0:
            self.carry = total
            return None, None
        if self.time_ldpc:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        bits, trials = self.ldpc.execute_dev(self.llr[:ready])
        if self.time_ldpc:
            e1.record()
            self.ldpc_events.append((e0, e1, ready))
        if self.outer_code:
            self.outer_code_status = self.bch.correct_dev(bits)
        out = self.bch.execute_dev(bits)
        rest = total - ready
        if rest:
            self.llr[:rest] = self.llr[ready:total].clone()
        self.carry = rest
        self.last_tags, self.tags = self.tags[:ready], self.tags[ready:]
        return out, trials

    # ---- the same in two halves for a two-stream software pipeline (whole frames, every FEC frame decoded: flush semantics):
    # stage_llr = time de-interleaver + demapper into LLR buffer `slot`; stage_fec = LDPC + descrambler from that buffer
    def stage_llr(self, F, slot):
        torch = self.torch
        if not hasattr(self, "llr2"):
            self.llr2 = [self.llr, torch.empty_like(self.llr)]
        assert self.carry == 0 and len(self.ti_blocks) == 1
        n_ti = self.num_blocks * self.cells_per_fec
        self.ti[0].l1_dyn(self.num_blocks)
        self.ti[0].execute_blocks_dev(self.cells[:F, :n_ti], self.ti_out[:F])
        for f in range(F):
            self.demap.execute_dev(self.ti_out[f], out=self.llr2[slot][f * self.num_blocks:(f + 1) * self.num_blocks])
        return F * self.num_blocks

    # finer stages for the four-stream schedule of receiver.pipeline_step (single LLR buffer: slot 0)
    def stage_ti_stats(self, F):
        torch = self.torch
        if not hasattr(self, "llr2"):
            self.llr2 = [self.llr, torch.empty_like(self.llr)]
        if not hasattr(self, "sums"):
            self.sums = torch.zeros((self.max_frames, 4), dtype=torch.float32, device=self.dev)
        assert len(self.ti_blocks) == 1, "the staged schedule covers one TI block per frame"
        n_ti = self.num_blocks * self.cells_per_fec
        self.ti[0].l1_dyn(self.num_blocks)
        self.ti[0].execute_blocks_dev(self.cells[:F, :n_ti], self.ti_out[:F])
        self.demap.stats_batch_dev(self.ti_out[:F, :n_ti], self.sums)

    def stage_llr_only(self, F, slot=0):
        n_ti = self.num_blocks * self.cells_per_fec
        if self.ti_out.shape[1] == n_ti:                 # TI blocks back to back: one launch for the whole buffer
            return self.demap.llr_batch_dev(self.ti_out, F, n_ti, self.sums, self.llr2[slot])
        for f in range(F):
            self.demap.llr_dev(self.ti_out[f], self.sums[f], self.llr2[slot][f * self.num_blocks:(f + 1) * self.num_blocks])
        return F * self.num_blocks

    def stage_fec(self, count, slot):
        torch = self.torch
        if self.time_ldpc:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        bits, trials = self.ldpc.execute_dev(self.llr2[slot][:count])
        if self.time_ldpc:
            e1.record()
            self.ldpc_events.append((e0, e1, count))
        if self.outer_code:
            self.outer_code_status = self.bch.correct_dev(bits)
        return self.bch.execute_dev(bits), trials

    def ts_from_bits(self, bits_host, trials_host, tags=None):
        """BBFRAME bits of decoded FEC frames -> TS bytes; SIMD batches the LDPC gave up on (-1) are dropped as the
        reference drops them (ldpc_decoder.cpp:264-268); frames of PLPs other than need_plp are skipped by the de-framer
        (bb_de_header.cpp:139-142). tags: PLP of every frame (default: those of the frames demod_cells_dev last returned)."""
        out = []
        buf = np.zeros(bits_host.shape[1] // 8 + 400, np.uint8)
        err = ctypes.c_int(0)
        if tags is None:
            tags = self.last_tags if len(self.last_tags) == bits_host.shape[0] else [self.need_plp] * bits_host.shape[0]
        for i in range(bits_host.shape[0]):
            if trials_host[i // self.group] < 0:
                continue
            n = self._l.t2gpu_bbdh_execute(self._bbdh, tags[i], bits_host.shape[1], bits_host[i].ctypes.data, buf.ctypes.data,
                                           buf.size, ctypes.byref(err))
            if n > 0:
                out.append(buf[:n].copy())
        return np.concatenate(out) if out else np.zeros(0, np.uint8)
