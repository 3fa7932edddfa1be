"""int16 IQ -> TS: the whole receive path of ``dvbt2_demodulator::execute`` over the C ABI, for buffers of whole T2 frames.

Call sequence per buffer (reference: src/DVB_T2/dvbt2_demodulator.cpp): front end (:145-226) -> P1 detection at every frame
start (p1_symbol::execute via symbol_acquisition :279-310) -> guard-interval correlation of every symbol (:321-330) -> FFT with
the guard dropped (:332-334) -> P2 / data / frame-closing equalisers -> time de-interleaver -> demapper -> LDPC -> BCH stub ->
BBFRAME de-framing. The tracking loops run OPEN here: the loop values are inputs (zeros and the nominal resample for a
synchronous source), the estimates the stages produce (P1 position and CFO, guard correlation, equaliser sync sums) are
returned to the caller instead of being fed back symbol by symbol -- the batch form of SURVEY.md section 8(e)."""
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
        self.p1 = p1_symbol(max_samples=max_frames * 4096, device=device)
        self.stream = torch.zeros(n_max + 64, dtype=torch.complex64, device=self.chain.dev)
        self.search = P1_LEN + 1024                                                    # samples searched from each frame start

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
        if np.abs(p2_start - (first + starts)).max() > 2:
            raise RuntimeError("frames are not equally spaced: P2 starts %s" % p2_start)
        o = self.chain.ofdm
        cp = torch.empty((n_frames * self.chain.n_sym, 4), dtype=torch.float32, device=self.chain.dev)
        from ._lib import lib, check
        check(lib().t2gpu_cp_correlate_stream_dev(self.stream.data_ptr(), first, self.frame_len, self.chain.n_sym, n_frames * self.chain.n_sym,
                                                  o.fft_size, o.guard_interval_size, cp.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "t2gpu_cp_correlate_stream_dev")
        bits, trials = self.chain.demod_stream_dev(self.stream, first + o.guard_interval_size, self.frame_len, n_frames, flush)
        return dict(bits=bits, trials=trials, p1=res, p2_start=p2_start, cp=cp.reshape(n_frames, self.chain.n_sym, 4))
