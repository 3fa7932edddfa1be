"""int16 IQ -> TS: the whole receive path of ``dvbt2_demodulator::execute`` over the C ABI, for buffers of whole T2 frames.

Call sequence per buffer (reference: src/DVB_T2/dvbt2_demodulator.cpp): front end (:145-226) -> P1 detection at every frame
start (p1_symbol::execute via symbol_acquisition :279-310) -> guard-interval correlation of every symbol (:321-330) -> FFT with
the guard dropped (:332-334) -> P2 / data / frame-closing equalisers -> time de-interleaver -> demapper -> LDPC -> BCH stub ->
BBFRAME de-framing. The tracking loops run OPEN here: the loop values are inputs (zeros and the nominal resample for a
synchronous source), the estimates the stages produce (P1 position and CFO, guard correlation, equaliser sync sums) are
returned to the caller instead of being fed back symbol by symbol -- the batch form of SURVEY.md section 8(e)."""
import ctypes

import numpy as np

from .chain import t2_chain
from .front import front_end, cp_correlate_dev, SAMPLE_RATE
from .p1 import p1_symbol

P1_LEN = 2048


class t2_receiver(object):
    def __init__(self, chain_args, chain_kwargs=None, max_frames=4, id_device=0, sample_rate=SAMPLE_RATE, device=0):
        import torch
        self.torch = torch
        kw = dict(chain_kwargs or {})
        kw.update(max_frames=max_frames, device=device)
        self.chain = t2_chain(*chain_args, **kw)
        o = self.chain.ofdm
        self.sym_size = o.fft_size + o.guard_interval_size
        self.frame_len = P1_LEN + self.chain.n_sym * self.sym_size                     # samples of one T2 frame at 64/7 MHz
        self.max_frames = max_frames
        n_max = max_frames * self.frame_len + 4096
        self.front = front_end(id_device=id_device, sample_rate=sample_rate, max_samples=n_max, device=device)
        self.p1 = p1_symbol(max_samples=max(max_frames * 4096, 2 * self.sym_size + 8192), device=device)
        self.stream = torch.zeros(n_max + 64, dtype=torch.complex64, device=self.chain.dev)
        self.search = P1_LEN + 1024                                                    # samples searched from each frame start
        # the batch form places every frame's FFT windows from the first frame's P1 position; the P1 arg-max jitters by a few
        # samples with noise, which the guard interval absorbs
        self.timing_slack = min(16, max(2, o.guard_interval_size // 8))

    def close(self):
        self.chain.close()
        self.front.close()
        self.p1.close()

    def demod_iq_dev(self, d_i, d_q, n_frames, level_detect=None, first_call=True, flush=False, loops=None):
        """d_i, d_q: int16 device tensors holding n_frames whole frames starting at a P1 symbol (stride 2 for AirSpy).
        Returns dict(bits, trials, p1 results, P2 start per frame, guard-correlation estimates [frames][symbols][4])."""
        torch = self.torch
        n_in = n_frames * self.frame_len
        pe, fe, rs = loops if loops is not None else (None, None, None)
        cells, _ = self.front.execute_dev(d_i, d_q, [n_in], self.stream, pe, fe, rs)
        assert cells == n_in, (cells, n_in)                                            # nominal resample: one cell per input sample
        if level_detect is None:
            level_detect = float(self.front.state()["level_detect"])                   # what execute() hands to p1_symbol (:235,283)
        starts = np.arange(n_frames, dtype=np.int64) * self.frame_len
        lens = np.minimum(self.search, n_in - starts).astype(np.int32)
        res, cons = self.p1.execute_batch_dev(self.stream, starts, lens, first_call, level_detect)
        p2_start = np.array([s + c - r.idx_buffer_sym if r.detected else -1 for s, c, r in zip(starts, cons, res)], np.int64)
        if (p2_start < 0).any():
            raise RuntimeError("P1 not found in %d of %d frames" % (int((p2_start < 0).sum()), n_frames))
        first = int(p2_start[0])
        if np.abs(p2_start - (first + starts)).max() > self.timing_slack:
            raise RuntimeError("frames are not equally spaced: P2 starts %s" % p2_start)
        o = self.chain.ofdm
        cp = torch.empty((n_frames * self.chain.n_sym, 4), dtype=torch.float32, device=self.chain.dev)
        from ._lib import lib, check
        check(lib().t2gpu_cp_correlate_stream_dev(self.stream.data_ptr(), first, self.frame_len, self.chain.n_sym, n_frames * self.chain.n_sym,
                                                  o.fft_size, o.guard_interval_size, cp.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "t2gpu_cp_correlate_stream_dev")
        bits, trials = self.chain.demod_stream_dev(self.stream, first + o.guard_interval_size, self.frame_len, n_frames, flush)
        return dict(bits=bits, trials=trials, p1=res, p2_start=p2_start, cp=cp.reshape(n_frames, self.chain.n_sym, 4))

    # ---- software pipeline over four HIP streams. The LDPC's persistent workgroups hold almost all LDS of every CU for the
    # whole decode, so of the other kernels only those with (nearly) no LDS can run beside it: the front end, P1, guard correlation,
    # equalisers, time de-interleaver and the demapper's statistics pass. The 32K FFT (128 KB of LDS per workgroup) and the
    # demapper's LLR pass (one FEC frame in LDS) get their own streams and fall into the gap between two decodes. Call k enqueues
    #   sa: front end, P1, guard correlation of buffer k;  equalisers, TI, demapper statistics of buffer k-1
    #   sf: FFT of buffer k          sd: LLR pass of buffer k-1          sb: LDPC + descrambler of buffer k-1
    # with events for the true dependencies; the result returned by call k belongs to buffer k-1 (None for the first call).
    # MEASURED (round 1, MI355X, CFG-A): running anything beside the decoder costs more than it hides. The 32 workgroups of a SIMD
    # batch meet at every sweep, so one workgroup slowed by a neighbour on its CU slows its whole batch: the decode goes from
    # 31.3 ms to 35-42 ms (10-step runs) while only ~3 ms of other work is hidden. The default is therefore T2GPU_PIPE_SERIAL=1:
    # same stages and streams, each call drained before the next. T2GPU_PIPE_SERIAL=0 still enqueues the overlapped schedule, but since
    # round 2 the front-end kernels hold 19-35 KB of LDS and cannot be co-resident with the decoder's workgroups: they queue behind it
    # (or, started first, keep part of the decoder's grid out until they finish). Kept for experiments only.
    def pipeline_step(self, d_i, d_q, n_frames, level_detect, first_call=False):
        torch = self.torch
        c, o = self.chain, self.chain.ofdm
        if not hasattr(self, "_pipe"):
            import os
            hi = int(os.environ.get("T2GPU_PIPE_LDPC_PRIORITY", "0"))           # lower number = higher priority; equal measured best
            mk = lambda prio=0: torch.cuda.Stream(device=c.dev, priority=prio)
            self._pipe = dict(sa=mk(), sf=mk(), sd=mk(), sb=mk(hi), k=0, serial=bool(int(os.environ.get("T2GPU_PIPE_SERIAL", "1"))), prev=None, fft_done=None, eq_done=None, llr_done=None,
                              ldpc_done=None,
                              spec=torch.empty((self.max_frames * c.n_sym, o.fft_size, 2), dtype=torch.float32, device=c.dev))
        pp = self._pipe
        sa, sf, sd, sb = pp["sa"], pp["sf"], pp["sd"], pp["sb"]
        ev = lambda st: (lambda e: (e.record(st), e)[1])(torch.cuda.Event())
        out = None
        cur = None
        with torch.cuda.stream(sa):
            if pp["ldpc_done"] is not None and not pp["serial"]:
                c.ldpc.wait_resident()                                           # beside the decoder, not ahead of it (see t2gpu.h)
            if d_i is not None:
                if pp["fft_done"] is not None:
                    sa.wait_event(pp["fft_done"])                                # the FFT that still reads the sample stream
                cur = self._front_p1_cp(d_i, d_q, n_frames, level_detect, first_call)
                front_done = ev(sa)
            prev = pp["prev"]
            if prev is not None:
                if pp["llr_done"] is not None:
                    sa.wait_event(pp["llr_done"])                                # the LLR pass that still reads ti_out / sums
                sa.wait_event(pp["fft_done"])
                c.spectrum_to_cells(pp["spec"][:prev["n"] * c.n_sym].reshape(prev["n"], c.n_sym, o.fft_size, 2))
                pp["eq_done"] = ev(sa)
                c.stage_ti_stats(prev["n"])
                stats_done = ev(sa)
        if cur is not None:
            with torch.cuda.stream(sf):
                sf.wait_event(front_done)
                if pp["eq_done"] is not None:
                    sf.wait_event(pp["eq_done"])                                 # the equaliser that still reads the spectrum buffer
                o.fft_stream_dev(self.stream, cur["first"] + o.guard_interval_size, self.frame_len, c.n_sym,
                                 o.fft_size + o.guard_interval_size, n_frames * c.n_sym, out=pp["spec"][:n_frames * c.n_sym])
                pp["fft_done"] = ev(sf)
        if prev is not None:
            with torch.cuda.stream(sd):
                sd.wait_event(stats_done)
                if pp["ldpc_done"] is not None:
                    sd.wait_event(pp["ldpc_done"])                               # the decode that still reads the LLR buffer
                count = c.stage_llr_only(prev["n"], 0)
                pp["llr_done"] = ev(sd)
            with torch.cuda.stream(sb):
                sb.wait_event(pp["llr_done"])
                bits, trials = c.stage_fec(count, 0)
                pp["ldpc_done"] = ev(sb)
            out = dict(prev["info"], bits=bits, trials=trials)
        pp["prev"] = None if cur is None else dict(n=n_frames, info=dict(p1=cur["p1"], p2_start=cur["p2_start"], cp=cur["cp"]))
        if pp["serial"]:
            self.pipeline_sync()                                                 # A/B baseline: same kernels, no overlap
        return out

    def pipeline_flush(self):
        """Decode the buffer still in flight (the last call's); returns its result."""
        return self.pipeline_step(None, None, 0, 0.0)

    def _front_p1_cp(self, d_i, d_q, n_frames, level_detect, first_call):
        """Front end, P1 windows, guard correlation of one buffer on the current stream (first part of demod_iq_dev)."""
        torch = self.torch
        n_in = n_frames * self.frame_len
        cells, _ = self.front.execute_dev(d_i, d_q, [n_in], self.stream)
        assert cells == n_in, (cells, n_in)
        starts = np.arange(n_frames, dtype=np.int64) * self.frame_len
        lens = np.minimum(self.search, n_in - starts).astype(np.int32)
        res, cons = self.p1.execute_batch_dev(self.stream, starts, lens, first_call, level_detect)
        p2_start = np.array([s + c - r.idx_buffer_sym if r.detected else -1 for s, c, r in zip(starts, cons, res)], np.int64)
        if (p2_start < 0).any():
            raise RuntimeError("P1 not found in %d of %d frames" % (int((p2_start < 0).sum()), n_frames))
        first = int(p2_start[0])
        if np.abs(p2_start - (first + starts)).max() > self.timing_slack:
            raise RuntimeError("frames are not equally spaced: P2 starts %s" % p2_start)
        o = self.chain.ofdm
        cp = torch.empty((n_frames * self.chain.n_sym, 4), dtype=torch.float32, device=self.chain.dev)
        from ._lib import lib, check
        check(lib().t2gpu_cp_correlate_stream_dev(self.stream.data_ptr(), first, self.frame_len, self.chain.n_sym, n_frames * self.chain.n_sym,
                                                  o.fft_size, o.guard_interval_size, cp.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "t2gpu_cp_correlate_stream_dev")
        return dict(first=first, p1=res, p2_start=p2_start, cp=cp.reshape(n_frames, self.chain.n_sym, 4))

    def pipeline_sync(self):
        if hasattr(self, "_pipe"):
            for k in ("sa", "sf", "sd", "sb"):
                self._pipe[k].synchronize()


class rx_config(ctypes.Structure):
    _fields_ = [("id_device", ctypes.c_int32), ("sample_rate", ctypes.c_float)] + \
               [(n, ctypes.c_int32) for n in ("fft_mode carrier_mode pilot_pattern guard_interval_mode papr_mode n_data l1_post_size plp_mod "
                                              "plp_fec_type plp_cod plp_rotation plp_num_blocks max_frames ldpc_group ldpc_trials "
                                              "saturate_llr").split()]


class rx_geometry(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in "frame_len n_sym fft_size guard_interval_size frame_cells fec_frames_per_t2_frame k_bch k_ldpc".split()]


class rx_ts_counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in "t2_frames l1_pre_crc_errors l1_post_crc_errors l1_mismatches fec_frames fec_frames_dropped_ldpc "
                                              "fec_frames_dropped_l1 bbheader_crc_errors ts_packet_errors resync ts_bytes ts_bytes_pending "
                                              "device_errors".split()]


class t2_rx(object):
    """The same batch receiver as ``t2_receiver`` with nothing but the C ABI underneath (``t2gpu_rx_*``, csrc/t2gpu_rx.cpp): buffers,
    stage sequencing and launches live in the library; Python hands over two device pointers per buffer. This is what bench.py times."""

    def __init__(self, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data, l1_post_size, plp_mod, plp_fec_type,
                 plp_cod, plp_rotation, plp_num_blocks, max_frames=4, ldpc_group=32, ldpc_trials=25, saturate_llr=False, id_device=0,
                 sample_rate=0.0, device=0):
        from ._lib import lib, T2GpuError
        self._l = lib()
        self.cfg = rx_config(id_device, sample_rate, fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data,
                             l1_post_size, plp_mod, plp_fec_type, plp_cod, plp_rotation, plp_num_blocks, max_frames, ldpc_group,
                             ldpc_trials, int(bool(saturate_llr)))
        self._h = self._l.t2gpu_rx_create(ctypes.byref(self.cfg), device)
        if not self._h:
            raise T2GpuError("t2gpu_rx_create: " + self._l.t2gpu_last_error().decode())
        self.geometry = rx_geometry()
        self._l.t2gpu_rx_info(self._h, ctypes.byref(self.geometry))
        self.frame_len, self.k_bch = self.geometry.frame_len, self.geometry.k_bch
        self.group = ldpc_group
        self.max_frames = max_frames

    def close(self):
        if getattr(self, "_h", None):
            self._l.t2gpu_rx_destroy(self._h)
            self._h = None

    __del__ = close

    def _stream(self, stream):
        import torch
        return torch.cuda.current_stream().cuda_stream if stream is None else stream

    def _check(self, rc, what):
        if rc < 0:
            from ._lib import T2GpuError
            raise T2GpuError(what + ": " + self._l.t2gpu_last_error().decode())
        return rc

    def front_dev(self, d_i, d_q, n_frames, level_detect=0.0, first_call=False, stream=None):
        return self._check(self._l.t2gpu_rx_front_dev(self._h, d_i.data_ptr(), d_q.data_ptr(), n_frames, float(level_detect), int(first_call),
                                                      self._stream(stream)), "t2gpu_rx_front_dev")

    def back_dev(self, n_frames, stream=None):
        return self._check(self._l.t2gpu_rx_back_dev(self._h, n_frames, None, None, self._stream(stream)), "t2gpu_rx_back_dev")

    def execute_dev(self, d_i, d_q, n_frames, level_detect=0.0, first_call=False, stream=None, flush=False):
        """d_i, d_q: int16 device tensors with n_frames whole frames from a P1 symbol on. Enqueues the whole path; returns the
        number of FEC frames decoded by this call: the complete SIMD batches of (frames waiting from earlier calls + this call's), as
        the reference forms them (llr_demapper.cpp:742-764); the rest waits in the handle (``carry``). flush=True (end of a stream)
        also decodes the waiting frames as one short batch, which the reference never does. Results: fetch / fetch_packed / results."""
        n = self._check(self._l.t2gpu_rx_execute_dev(self._h, d_i.data_ptr(), d_q.data_ptr(), n_frames, float(level_detect),
                                                     int(first_call), None, None, self._stream(stream)), "t2gpu_rx_execute_dev")
        return n + self.flush_dev(stream) if flush else n

    def flush_dev(self, stream=None):
        return self._check(self._l.t2gpu_rx_flush_dev(self._h, self._stream(stream)), "t2gpu_rx_flush_dev")

    @property
    def carry(self):
        return self._l.t2gpu_rx_carry(self._h)

    def set_overlap(self, enable=True):
        """The decode of a call on the handle's own stream, beside the next call's front half (t2gpu_rx_set_overlap): for calls of one or
        two T2 frames, whose SIMD batches leave CUs free. Results are complete after wait() / any fetch."""
        self._check(self._l.t2gpu_rx_set_overlap(self._h, int(bool(enable))), "t2gpu_rx_set_overlap")

    def wait(self):
        self._check(self._l.t2gpu_rx_wait(self._h), "t2gpu_rx_wait")

    def reset(self):
        self._check(self._l.t2gpu_rx_reset(self._h), "t2gpu_rx_reset")

    def ldpc_occupancy(self):
        v = (ctypes.c_int * 6)()
        self._check(self._l.t2gpu_rx_ldpc_occupancy(self._h, v), "t2gpu_rx_ldpc_occupancy")
        return dict(zip(("workgroups_per_cu", "waves_per_workgroup", "lds_bytes_per_workgroup", "frames_per_workgroup", "cus", "batches_resident"),
                        (int(x) for x in v)))

    def fetch(self, count):
        """(bits uint8 [count][k_bch] one bit per byte, trials-left int32 per SIMD batch) of the last back half (+ flush)."""
        bits = np.empty((count, self.k_bch), np.uint8)
        trials = np.empty(((count + self.group - 1) // self.group,), np.int32)
        self._check(self._l.t2gpu_rx_fetch(self._h, count, bits.ctypes.data, trials.ctypes.data), "t2gpu_rx_fetch")
        return bits, trials

    def fetch_packed(self, count):
        """(bytes uint8 [count][k_bch / 8] packed MSB first, trials-left int32 per SIMD batch): what crosses the bus."""
        rows = np.empty((count, self.k_bch // 8), np.uint8)
        trials = np.empty(((count + self.group - 1) // self.group,), np.int32)
        self._check(self._l.t2gpu_rx_fetch_packed(self._h, count, rows.ctypes.data, trials.ctypes.data), "t2gpu_rx_fetch_packed")
        return rows, trials

    # ---- the host end inside the library: per-frame L1 parse + BBFRAME de-framing on a worker thread, overlapped with later calls
    def ts_enable(self, need_plp=0, l1_check=True):
        self._check(self._l.t2gpu_rx_ts_enable(self._h, need_plp, int(bool(l1_check))), "t2gpu_rx_ts_enable")

    def ts_read(self, wait_all=True):
        """All TS bytes the worker has finished (wait_all: after every call enqueued so far has been de-framed), oldest first."""
        c = rx_ts_counters()
        self._check(self._l.t2gpu_rx_ts_counters_get(self._h, int(wait_all), ctypes.byref(c)), "t2gpu_rx_ts_counters_get")
        out = np.empty(int(c.ts_bytes_pending), np.uint8)
        n = self._check(self._l.t2gpu_rx_ts_read(self._h, out.ctypes.data, out.size, 0), "t2gpu_rx_ts_read") if out.size else 0
        return out[:n]

    def ts_read_into(self, buf, wait_all=False):
        """The same into a caller's uint8 buffer (no allocation, no page faults in a consumer loop); returns the number of bytes written
        (<= buf.size; what does not fit stays queued)."""
        return self._check(self._l.t2gpu_rx_ts_read(self._h, buf.ctypes.data, buf.size, int(wait_all)), "t2gpu_rx_ts_read")

    def ts_counters(self, wait_all=True):
        c = rx_ts_counters()
        self._check(self._l.t2gpu_rx_ts_counters_get(self._h, int(wait_all), ctypes.byref(c)), "t2gpu_rx_ts_counters_get")
        return {n: int(getattr(c, n)) for n, _ in rx_ts_counters._fields_}

    def set_outer_code(self, enable=True):
        """Opt-in BCH check / correction of the LDPC output before the descrambler (the reference does none, bch_decoder.cpp:136)."""
        self._check(self._l.t2gpu_rx_set_outer_code(self._h, int(enable)), "t2gpu_rx_set_outer_code")

    def outer_code_status(self, count):
        st = np.empty((count,), np.int32)
        self._check(self._l.t2gpu_rx_outer_code_status(self._h, count, st.ctypes.data), "t2gpu_rx_outer_code_status")
        return st

    STAGES = ("front", "p1", "guard_corr", "fft", "equalise", "ti", "demap", "ldpc", "descramble")

    def stage_ms(self):
        """Durations (ms) of the stages of the last call from HIP events on its stream; waits for it. -1: not run."""
        ms = (ctypes.c_float * 9)()
        self._check(self._l.t2gpu_rx_stage_ms(self._h, ms), "t2gpu_rx_stage_ms")
        return dict(zip(self.STAGES, (float(v) for v in ms)))

    def fft_eq_demap_dev(self, n_frames, stream=None):
        """BASELINE config 2 on the frames of the last front half: FFT + equalisers + time de-interleave + demap. Enqueue only."""
        return self._check(self._l.t2gpu_rx_fft_eq_demap_dev(self._h, n_frames, self._stream(stream)), "t2gpu_rx_fft_eq_demap_dev")

    def sync_sums(self, n_frames):
        out = np.zeros((n_frames * self.geometry.n_sym, 2), np.float32)
        self._check(self._l.t2gpu_rx_sync_sums(self._h, n_frames, out.ctypes.data), "t2gpu_rx_sync_sums")
        return out

    def last_ldpc_ms(self):
        """Duration of the last LDPC launch (HIP events on the stream it ran on); waits for it."""
        ms = ctypes.c_float(0)
        self._check(self._l.t2gpu_rx_results(self._h, 0, None, None, None, None, ctypes.byref(ms)), "t2gpu_rx_results")
        return ms.value

    def results(self, n_frames):
        from .p1 import p1_result
        p1 = (p1_result * n_frames)()
        p2 = np.zeros(n_frames, np.int64)
        cp = np.zeros((n_frames, self.geometry.n_sym, 4), np.float32)
        level, ms = ctypes.c_float(0), ctypes.c_float(0)
        self._check(self._l.t2gpu_rx_results(self._h, n_frames, p1, p2.ctypes.data, cp.ctypes.data, ctypes.byref(level), ctypes.byref(ms)),
                    "t2gpu_rx_results")
        return dict(p1=list(p1), p2_start=p2, cp=cp, level_detect=level.value, ldpc_ms=ms.value)


class t2_closed_loop(object):
    """The reference's own operating mode: ``dvbt2_demodulator::execute`` + ``symbol_acquisition`` symbol by symbol, every
    tracking loop closed (src/DVB_T2/dvbt2_demodulator.cpp:145-254,267-448) -- chunk sizing from est_chunk (:151-163), P1 search
    with carried correlator state, guard-interval frequency loop once L1-pre has passed its CRC (:321-330), phase PI loop and
    the bang-bang sample-rate tracker from the equalisers' pilot sums (:429-439), Farrow ratio and NCO values fed back into the
    next chunk. Compute is the same C-ABI stages as the batch receiver, one symbol per call; the state machine is host code as in
    the reference. The tuner the reference re-tunes when P1 reports >= 10 Hz (:291-305) is emulated by an extra NCO term."""

    def __init__(self, rx):
        self.rx = rx
        self.torch = rx.torch
        from .front import sync_loops
        self.sync = sync_loops()
        c, o = rx.chain, rx.chain.ofdm
        self.sym_size = rx.sym_size
        dev = c.dev
        self.buffer_sym = self.torch.zeros(self.sym_size + 8, dtype=self.torch.complex64, device=dev)
        self.out = self.torch.zeros(2 * (self.sym_size + 4096), dtype=self.torch.complex64, device=dev)
        self.spec = self.torch.zeros((1, o.fft_size, 2), dtype=self.torch.float32, device=dev)
        self.next_symbol_type = "P1"
        self.est_chunk = 0
        self.idx_buffer_sym = 0
        self.idx_symbol = 0
        self.crc32_l1_pre = False
        self.tuner = 0.0                      # rad/sample, the emulated re-tune
        self.cell_pos = 0
        self.log = []                         # (symbol type, frequency_est_filtered, phase_est_filtered, resample) per symbol
        self.p1_seen = 0
        self.l1 = None
        self.level_detect = None              # set by the caller when the AGC has settled (signal_->gain_changed, :283); else the
                                              # front end's own estimate of the previous buffer is used
        rx.front.reset()
        rx.p1.reset()

    def execute(self, d_i, d_q, level_gain_changed=True):
        """d_i, d_q: int16 device tensors (one execute() buffer). Returns the list of (bits, trials) of the frames completed."""
        torch = self.torch
        rx, c, o = self.rx, self.rx.chain, self.rx.chain.ofdm
        from .l1 import l1_pre_info, l1_post_info
        len_in = d_i.numel() // rx.front.stride
        idx_in, done = 0, []
        level = self.level_detect if self.level_detect is not None else float(rx.front.state()["level_detect"])
        while idx_in < len_in:
            if self.est_chunk == 0:                                                      # :151-155
                self.est_chunk = (2048 if self.next_symbol_type == "P1" else 0) + self.sym_size
            g = self.sync.get()
            resample = g["arbitrary_resample"]
            chunk = min(int(np.rint(self.est_chunk * resample * 2)), len_in - idx_in)    # :160-162
            s = rx.front.stride
            n_out, _ = rx.front.execute_dev(d_i[idx_in * s:], d_q[idx_in * s:], [chunk], self.out, [g["phase_est_filtered"]],
                                            [np.float32(g["frequency_est_filtered"] + np.float32(self.tuner))], [resample])
            idx_in += chunk
            # ---- symbol_acquisition (:267-448)
            consume = 0
            while consume < n_out:
                if self.next_symbol_type == "P1":
                    det, consume, r = rx.p1.execute_dev(self.out[:n_out], consume, level_gain_changed, level)
                    if det:
                        self.p1_seen += 1
                        k = r.idx_buffer_sym
                        self.buffer_sym[:k] = self.out[consume - k:consume]              # p1_symbol.cpp:97
                        self.idx_buffer_sym = k
                        if abs(r.coarse_freq_offset) >= 10.0 and not self.crc32_l1_pre:  # :291-305: ask the tuner, wait for the next P1
                            self.tuner += 2.0 * np.pi * r.coarse_freq_offset / (64.0e6 / 7.0)
                            self.idx_buffer_sym = 0
                        else:
                            self.next_symbol_type = "P2"
                    continue
                n = min(n_out - consume, self.sym_size - self.idx_buffer_sym)
                self.buffer_sym[self.idx_buffer_sym:self.idx_buffer_sym + n] = self.out[consume:consume + n]
                consume += n
                self.idx_buffer_sym += n
                if self.idx_buffer_sym < self.sym_size:
                    self.est_chunk = self.sym_size - self.idx_buffer_sym                 # :339
                    continue
                self.idx_buffer_sym = 0
                if self.crc32_l1_pre:                                                    # :321-330
                    cp = cp_correlate_dev(self.buffer_sym[:self.sym_size].reshape(1, -1), o.fft_size, o.guard_interval_size)
                    self.sync.frequency(float(cp[0, 2]), o.fft_size)
                o.fft_stream_dev(self.buffer_sym, o.guard_interval_size, 0, 1, self.sym_size, 1, out=self.spec)   # :332-334
                self.est_chunk = 0
                kind = self.next_symbol_type
                if kind == "P2":
                    self.idx_symbol = 0
                    cells, sync = o.eq_p2_dev(self.spec)
                    host = cells[0].cpu().numpy().view(np.complex64).reshape(-1)
                    ok, pre = l1_pre_info(host)
                    self.crc32_l1_pre = bool(ok)
                    if ok:
                        okp, post, plp, dyn = l1_post_info(host, pre)
                        self.l1 = (pre, post, plp, dyn) if okp else None
                    c.cells[0, :o.c_p2 - c.p2_skip] = cells[0, c.p2_skip:]
                    self.cell_pos = o.c_p2 - c.p2_skip
                    self.idx_symbol = 1
                    self.next_symbol_type = "DATA"
                elif kind == "DATA":
                    cells, sync = o.eq_data_dev(self.spec, torch.tensor([self.idx_symbol], dtype=torch.int32, device=c.dev))
                    c.cells[0, self.cell_pos:self.cell_pos + o.c_data] = cells[0]
                    self.cell_pos += o.c_data
                    self.idx_symbol += 1
                    if self.idx_symbol == o.n_p2 + c.n_dat:
                        self.next_symbol_type = "FC" if o.l_fc else "P1"
                else:
                    cells, sync = o.eq_fc_dev(self.spec)
                    c.cells[0, self.cell_pos:self.cell_pos + o.n_fc] = cells[0]
                    self.cell_pos += o.n_fc
                    self.next_symbol_type = "P1"
                sv = sync[0].cpu().numpy()
                self.sync.symbol(float(sv[0]), float(sv[1]))                              # :429-439
                g2 = self.sync.get()
                self.log.append((kind, float(g2["frequency_est_filtered"]), float(g2["phase_est_filtered"]), g2["arbitrary_resample"]))
                if self.next_symbol_type == "P1" and kind != "P1":                       # frame complete: hand the cells on (emit data)
                    if self.crc32_l1_pre:
                        bits, trials = c.demod_cells_dev(1, flush=True)
                        done.append((bits.clone(), trials.clone()))                      # the chain reuses its output buffers
        return done
