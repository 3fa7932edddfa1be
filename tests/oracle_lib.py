"""Test-side bindings of the CPU checker: oracle/liboracle.so (our C restatement), oracle/_ref/libref_ldpc.so
(the reference's own LDPC headers compiled by oracle/Makefile, absent when never built) and the schedule emulator
tests/emu (host replay of the kernel's schedule). TEST INFRASTRUCTURE: nothing in the product imports this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_i8p = ctypes.POINTER(ctypes.c_int8)
_u8p = ctypes.POINTER(ctypes.c_uint8)

_cache = {}


def _make_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def oracle():
    if "ora" not in _cache:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            _make_oracle()
        _cache["ora"] = ctypes.CDLL(path)
    return _cache["ora"]


def ref():
    """The compiled reference LDPC (or None when oracle/_ref was never built on this machine)."""
    if "ref" not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "libref_ldpc.so")
        try:
            _cache["ref"] = ctypes.CDLL(path) if os.path.exists(path) else None
        except OSError:
            _cache["ref"] = None
    return _cache["ref"]


def emu():
    if "emu" not in _cache:
        bdir = os.path.join(ROOT, "tests", "_build")
        os.makedirs(bdir, exist_ok=True)
        path = os.path.join(bdir, "libemu.so")
        srcs = [os.path.join(ROOT, "tests", "emu", "ldpc_emu.cpp"),
                os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "csrc", "ldpc_graph.cpp")]
        deps = srcs + [os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "csrc", "ldpc_cn.h")]
        if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", path] + srcs)
        _cache["emu"] = ctypes.CDLL(path)
    return _cache["emu"]


def code_id(fec_type, code_rate):
    return fec_type * 6 + code_rate


def ldpc_params(cid):
    n, k, q, lt = (ctypes.c_int() for _ in range(4))
    assert oracle().ora_ldpc_params(cid, ctypes.byref(n), ctypes.byref(k), ctypes.byref(q), ctypes.byref(lt)) == 0
    return n.value, k.value, q.value, lt.value


def ldpc_encode(cid, info):
    n, k, _, _ = ldpc_params(cid)
    info = np.ascontiguousarray(info, dtype=np.uint8).reshape(-1, k)
    cw = np.zeros((info.shape[0], n), dtype=np.uint8)
    for b in range(info.shape[0]):
        oracle().ora_ldpc_encode(cid, info[b].ctypes.data_as(_u8p), cw[b].ctypes.data_as(_u8p))
    return cw


def make_llr(cid, frames, sigma, seed, scale=8.0, info=None):
    """Random codewords (or the codewords of the given information bits) through a BPSK/AWGN channel, quantised to the int8 LLR
    format the demapper produces (positive = bit 0). Returns (info bits, int8 LLRs)."""
    n, k, _, _ = ldpc_params(cid)
    rng = np.random.Generator(np.random.PCG64(seed))
    rnd = rng.integers(0, 2, size=(frames, k), dtype=np.uint8)
    info = rnd if info is None else np.ascontiguousarray(info, dtype=np.uint8)
    cw = ldpc_encode(cid, info)
    y = (1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.shape)
    llr = np.clip(np.rint(y * scale), -127, 127).astype(np.int8)
    return info, llr


def _decode(fn, cid, llr, trials):
    n, k, _, _ = ldpc_params(cid)
    llr = np.ascontiguousarray(llr, dtype=np.int8).reshape(-1, n)
    blocks = llr.shape[0]
    bits = np.full((blocks, k), 255, dtype=np.uint8)
    lo = np.zeros((blocks, n), dtype=np.int8)
    r = fn(cid, llr.ctypes.data_as(_i8p), blocks, trials, bits.ctypes.data_as(_u8p), lo.ctypes.data_as(_i8p))
    return r, bits, lo


def ora_decode(cid, llr, trials=25):
    """One reference-style batch (all frames stop together). Returns (trials_left, bits[blocks][k], llr_out)."""
    return _decode(oracle().ora_ldpc_decode, cid, llr, trials)


def ref_decode(cid, llr, trials=25):
    return _decode(ref().ref_ldpc_decode, cid, llr, trials)


def emu_decode(cid, llr, trials=25, order_mode=0):
    """Host replay of the kernel schedule; order_mode picks the thread order inside every barrier epoch
    (0 ascending, 1 descending, >= 2 seeded shuffles)."""
    n, k, _, _ = ldpc_params(cid)
    llr = np.ascontiguousarray(llr, dtype=np.int8).reshape(-1, n)
    blocks = llr.shape[0]
    bits = np.full((blocks, k), 255, dtype=np.uint8)
    lo = np.zeros((blocks, n), dtype=np.int8)
    r = emu().emu_ldpc_decode(cid, llr.ctypes.data_as(_i8p), blocks, trials, bits.ctypes.data_as(_u8p),
                              lo.ctypes.data_as(_i8p), order_mode)
    return r, bits, lo


def ora_decode_batched(cid, llr, group=32, trials=25):
    """Decode [frames][n] in reference batches of `group`; returns (trials_left[batches], bits, llr_out)."""
    n, k, _, _ = ldpc_params(cid)
    llr = np.ascontiguousarray(llr, dtype=np.int8).reshape(-1, n)
    frames = llr.shape[0]
    tl, bits, lo = [], np.zeros((frames, k), np.uint8), np.zeros((frames, n), np.int8)
    for b0 in range(0, frames, group):
        r, b, l = ora_decode(cid, llr[b0:b0 + group], trials)
        tl.append(r)
        lo[b0:b0 + group] = l
        bits[b0:b0 + group] = (l[:, :k] < 0)
    return np.array(tl, np.int32), bits, lo


# ------------------------------------------------------------------------------------------------ FEC-side stages
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)


def ora_bb_prbs(n):
    out = np.zeros(n, np.uint8)
    oracle().ora_bb_prbs(out.ctypes.data_as(_u8p), n)
    return out


def ora_bch_descramble(cid, bits):
    n, k, _, _ = ldpc_params(cid)
    bits = np.ascontiguousarray(bits, np.uint8).reshape(-1, k)
    kb = [7032, 9552, 10632, 11712, 12432, 13152, 32208, 38688, 43040, 48408, 51648, 53840][cid]
    out = np.zeros((bits.shape[0], kb), np.uint8)
    assert oracle().ora_bch_descramble(cid, bits.ctypes.data_as(_u8p), bits.shape[0], out.ctypes.data_as(_u8p)) == kb
    return out


def bch_params(cid):
    """(m, t, k_bch, n_bch) of LDPC code cid: parity bits = n_bch - k_bch = m * t (bch_decoder.cpp:79-134 holds the pairs)."""
    _, k, _, _ = ldpc_params(cid)
    kb = [7032, 9552, 10632, 11712, 12432, 13152, 32208, 38688, 43040, 48408, 51648, 53840][cid]
    m = 14 if cid < 6 else 16
    return m, (k - kb) // m, kb, k


def ora_bch_minpoly(m, j):
    o = oracle()
    o.ora_bch_minpoly.restype = ctypes.c_uint32
    return int(o.ora_bch_minpoly(m, j))


def ora_bch_generator(m, t):
    g = np.zeros(m * t + 1, np.uint8)
    deg = oracle().ora_bch_generator(m, t, g.ctypes.data_as(_u8p))
    return g[:deg + 1]


def ora_bch_encode(cid, msg):
    """[f][k_bch] -> [f][n_bch] systematic codewords (bitwise LFSR)."""
    m, t, kb, nb = bch_params(cid)
    out = np.zeros((msg.shape[0], nb), np.uint8)
    out[:, :kb] = msg
    for row in out:
        assert oracle().ora_bch_encode(m, t, row.ctypes.data_as(_u8p), kb, nb) == 0
    return out


def ora_bch_decode(cid, words):
    """[f][n_bch] received words -> (corrected copy, status per frame: corrected bits, or -1 = more than t errors)."""
    m, t, _, nb = bch_params(cid)
    out = np.ascontiguousarray(words, dtype=np.uint8).copy()
    st = np.zeros(out.shape[0], np.int32)
    for i, row in enumerate(out):
        st[i] = oracle().ora_bch_decode(m, t, row.ctypes.data_as(_u8p), nb)
    return out, st


def ora_bitdeint_address(mod, fec_type, code_rate):
    size = 64800 if fec_type == 1 else 16200
    a = np.zeros(size, np.int32)
    assert oracle().ora_bitdeint_address(mod, fec_type, code_rate, a.ctypes.data_as(_ip)) == size
    return a


def ora_demap(mod, fec_type, code_rate, rotation, cells, precision_override=0.0):
    """cells complex64 [n]; returns (llr [frames][fec_size], sums[3], derotated cells)."""
    c = np.ascontiguousarray(cells, np.complex64).copy()
    size = 64800 if fec_type == 1 else 16200
    cpf = size // (2 * (mod + 1))
    frames = c.size // cpf
    out = np.zeros((frames, size), np.int8)
    sums = np.zeros(3, np.float32)
    fn = oracle().ora_demap
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float]
    r = fn(mod, fec_type, code_rate, rotation, c.ctypes.data, c.size, out.ctypes.data, sums.ctypes.data, precision_override)
    assert r == frames
    return out, sums, c


def ora_cell_perm(num_blocks, cells_per_fec):
    p = np.zeros(num_blocks * cells_per_fec, np.int32)
    oracle().ora_cell_perm(num_blocks, cells_per_fec, p.ctypes.data_as(_ip))
    return p


def ora_ti_frame_walk(plps, dyns, num_cells, plp_state=0, max_out=4096):
    """plps: [(mod, fec_type, time_il_length, time_il_type)], dyns: [(id, start, num_blocks)] -> ([(plp, first, blocks, size)], state)"""
    o = oracle()
    P = np.ascontiguousarray(np.array(plps, np.int32).reshape(len(plps), 4))
    D = np.ascontiguousarray(np.array(dyns, np.int32).reshape(len(plps), 3))
    out = np.zeros((max_out, 4), np.int32)
    st = ctypes.c_int(plp_state)
    o.ora_ti_frame_walk.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int]
    rc = o.ora_ti_frame_walk(len(plps), P.ctypes.data, D.ctypes.data, num_cells, ctypes.byref(st), out.ctypes.data, max_out)
    if rc < 0:
        return rc, st.value
    return [tuple(int(v) for v in out[k]) for k in range(rc)], st.value


class OraTi(object):
    def __init__(self, cells_per_fec, num_blocks_max):
        o = oracle()
        o.ora_ti_create.restype = ctypes.c_void_p
        o.ora_ti_begin.argtypes = [ctypes.c_void_p, ctypes.c_int]
        o.ora_ti_push.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        o.ora_ti_destroy.argtypes = [ctypes.c_void_p]
        self._o, self._h = o, o.ora_ti_create(cells_per_fec, num_blocks_max)

    def begin(self, num_blocks):
        self._o.ora_ti_begin(self._h, num_blocks)

    def push(self, cells, out):
        cells = np.ascontiguousarray(cells, np.complex64)
        return self._o.ora_ti_push(self._h, cells.ctypes.data, cells.size, out.ctypes.data)

    def __del__(self):
        self._o.ora_ti_destroy(self._h)


# ------------------------------------------------------------------------------------------------ OFDM side
class OraMode(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("fft_mode", "carrier_mode", "pilot_pattern", "guard_interval_mode", "papr_mode", "n_data",
                                            "is32k", "fft_size", "k_total", "k_ext", "k_offset", "l_nulls", "n_p2", "c_p2", "c_data",
                                            "n_fc", "c_fc", "l_fc", "len_frame", "dx", "dy")] + \
               [(n, ctypes.c_float) for n in ("amp_sp", "amp_cp", "amp_p2")]


def ora_mode(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data):
    m = OraMode(fft_mode, carrier_mode, pilot_pattern, guard_interval_mode, papr_mode, n_data)
    rc = oracle().ora_mode_init(ctypes.byref(m))
    return m if rc == 0 else None


def ora_symbol_carriers(m, idx_symbol):
    mp = np.zeros(m.k_total, np.int32)
    rf = np.zeros(m.k_total, np.float32)
    oracle().ora_symbol_carriers(ctypes.byref(m), idx_symbol, mp.ctypes.data_as(_ip), rf.ctypes.data_as(_fp))
    return mp, rf


def ora_freq_deint(m, kind):
    cells = [m.c_p2, m.c_data, m.n_fc][kind]
    he, ho = np.zeros(cells, np.int32), np.zeros(cells, np.int32)
    assert oracle().ora_freq_deint(ctypes.byref(m), kind, he.ctypes.data_as(_ip), ho.ctypes.data_as(_ip)) == cells
    return he, ho


def ora_data_symbol(m, idx_symbol, ofdm_cell):
    """data_symbol::execute on one fft-shifted symbol (complex64[fft_size]) -> (cells complex64[c_data], phase_offset, sro)."""
    mp, rf = ora_symbol_carriers(m, idx_symbol)
    kind = 0 if idx_symbol < m.n_p2 else (2 if (m.l_fc and idx_symbol == m.len_frame - 1) else 1)
    he, ho = ora_freq_deint(m, kind)
    h = ho if idx_symbol % 2 == 0 else he
    x = np.ascontiguousarray(ofdm_cell, np.complex64)
    ncell = [m.c_p2, m.c_data, m.n_fc][kind]
    out = np.zeros(ncell, np.complex64)
    sync = np.zeros(2, np.float32)
    fn = oracle().ora_symbol_equalise
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
    n = fn(ctypes.addressof(m), kind, x.ctypes.data, mp.ctypes.data, rf.ctypes.data, h.ctypes.data, out.ctypes.data, sync.ctypes.data)
    assert n == ncell, (n, ncell)
    return out, float(sync[0]), float(sync[1])


# ---- sample-rate front end (oracle/front_oracle.c) and the reference's Qt-free DSP classes (oracle/_ref/libref_dsp.so) ----
def ref_dsp(strict=False):
    """The reference's Qt-free DSP headers compiled unmodified: with the reference's own flags (-Ofast), or strict=True
    without fast-math so every float operation happens in the order the reference source writes it."""
    key = "ref_dsp_strict" if strict else "ref_dsp"
    if key not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "lib%s.so" % key)
        try:
            _cache[key] = ctypes.CDLL(path) if os.path.exists(path) else None
        except OSError:
            _cache[key] = None
    return _cache[key]


def _c64(x):
    return np.ascontiguousarray(x, np.complex64)


class OraDecim(object):
    """filter_decimator restated; ref=True drives the reference class itself."""

    def __init__(self, ref=False, strict=False, scalar=False):
        self.lib = ref_dsp(strict) if ref else oracle()
        self.new, self.free, self.run = ((self.lib.ref_decim_new, self.lib.ref_decim_free, self.lib.ref_decim_execute) if ref else
                                         (self.lib.ora_decim_create, self.lib.ora_decim_destroy,
                                          self.lib.ora_decim_execute_scalar if scalar else self.lib.ora_decim_execute))
        self.new.restype = ctypes.c_void_p
        self.free.argtypes = [ctypes.c_void_p]
        self.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.h = self.new()

    def __call__(self, x, out=None):
        """out: a caller's complex64 buffer of len(x) // 2 + 1 cells at least (the result is then a view of it, no copy)."""
        x = _c64(x)
        if out is not None:
            return out[:self.run(self.h, len(x), x.ctypes.data, out.ctypes.data)]
        out = np.zeros(len(x) // 2 + 1, np.complex64)
        n = self.run(self.h, len(x), x.ctypes.data, out.ctypes.data)
        return out[:n].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.free(self.h)
            self.h = None


class OraFft(object):
    """fast_fourier_transform::execute restated (oracle/fft_oracle.c): forward transform of n cells, halves swapped."""

    def __init__(self, n):
        L = oracle()
        L.ora_fft_create.restype = ctypes.c_void_p
        L.ora_fft_create.argtypes = [ctypes.c_int]
        L.ora_fft_execute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ora_fft_destroy.argtypes = [ctypes.c_void_p]
        self.L, self.n = L, n
        self.h = L.ora_fft_create(n)
        assert self.h, "fft size must be a power of two"

    def __call__(self, x, shift=True):
        x = _c64(x)
        assert x.size == self.n
        out = np.empty(self.n, np.complex64)
        self.L.ora_fft_execute(self.h, x.ctypes.data, out.ctypes.data, int(shift))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_fft_destroy(self.h)
            self.h = None


class OraFarrow(object):
    """interpolator_farrow<complex,float> restated; ref=True drives the reference template itself."""

    def __init__(self, ref=False, strict=False):
        self.ref = ref
        self.lib = ref_dsp(strict) if ref else oracle()
        self.new, self.free, self.run = ((self.lib.ref_farrow_new, self.lib.ref_farrow_free, self.lib.ref_farrow_execute) if ref else
                                         (self.lib.ora_farrow_create, self.lib.ora_farrow_destroy, self.lib.ora_farrow_execute))
        self.new.restype = ctypes.c_void_p
        self.free.argtypes = [ctypes.c_void_p]
        self.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p] + ([] if ref else [ctypes.c_void_p])
        self.h = self.new()

    def __call__(self, x, resample, want_phases=False, out=None):
        """out: a caller's complex64 buffer of len(x) / resample + 8 cells at least (result = a view of it; not with want_phases)."""
        x = _c64(x)
        cap = int(len(x) / max(resample, 0.05)) + 8
        if out is not None and not want_phases:
            assert out.size >= cap
            return out[:self.run(self.h, len(x), x.ctypes.data, resample, out.ctypes.data, *(() if self.ref else (None,)))]
        out = np.zeros(cap, np.complex64)
        if self.ref:
            n = self.run(self.h, len(x), x.ctypes.data, resample, out.ctypes.data)
            return out[:n].copy()
        ph = np.zeros(cap, np.float32)
        n = self.run(self.h, len(x), x.ctypes.data, resample, out.ctypes.data, ph.ctypes.data)
        return (out[:n].copy(), ph[:n].copy()) if want_phases else out[:n].copy()

    def phase(self):
        """x1 after the last call (restated class only)."""
        fn = self.lib.ora_farrow_phase
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_void_p]
        return np.float32(fn(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.free(self.h)
            self.h = None


class OraPi(ctypes.Structure):
    _fields_ = [("k_p", ctypes.c_float), ("k_i", ctypes.c_float), ("old_integral", ctypes.c_float)]


class OraFront(object):
    """dvbt2_demodulator::execute front loop (dc / iq imbalance / NCO), chunk by chunk."""

    def __init__(self, id_device=0):
        o = oracle()
        o.ora_front_create.restype = ctypes.c_void_p
        o.ora_front_destroy.argtypes = [ctypes.c_void_p]
        o.ora_front_chunk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        o.ora_front_finish.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        o.ora_front_get_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.o = o
        self.h = o.ora_front_create(id_device)
        self.stride = 2 if id_device == 1 else 1

    def execute(self, i_in, q_in, chunk_len, phase_est_filtered, frequency_est_filtered, out=None):
        """One execute() call: chunks in order with their loop values. Returns the derotated stream (in `out` when given)."""
        i_in = np.ascontiguousarray(i_in, np.int16)
        q_in = np.ascontiguousarray(q_in, np.int16)
        total = int(np.sum(chunk_len))
        out = np.zeros(total, np.complex64) if out is None else out[:total]
        theta = np.zeros(3, np.float32)
        pos = 0
        for n, pe, fe in zip(chunk_len, phase_est_filtered, frequency_est_filtered):
            self.o.ora_front_chunk(self.h, int(n), i_in.ctypes.data, q_in.ctypes.data, pos, float(pe), float(fe),
                                   theta.ctypes.data, out[pos:].ctypes.data)
            pos += int(n)
        self.o.ora_front_finish(self.h, total, theta.ctypes.data)
        return out, theta

    def set_iq(self, c1, c2):
        self.o.ora_front_set_iq.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
        self.o.ora_front_set_iq(self.h, float(c1), float(c2))

    def state(self):
        v = np.zeros(8, np.float32)
        self.o.ora_front_get_state(self.h, v.ctypes.data)
        return dict(dc_re=v[0], dc_im=v[1], c1=v[2], c2=v[3], phase_nco=v[4], frequency_nco=v[5], level_detect=v[6])

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_front_destroy(self.h)
            self.h = None


def ora_cp_frequency_est(sym, fft_size, guard):
    sym = _c64(sym)
    s2 = np.zeros(2, np.float32)
    fn = oracle().ora_cp_frequency_est
    fn.restype = ctypes.c_float
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return float(fn(sym.ctypes.data, fft_size, guard, s2.ctypes.data)), complex(s2[0], s2[1])


class OraSync(object):
    """Tracking updates of symbol_acquisition (PI loop filters + the bang-bang sample-rate tracker)."""

    def __init__(self, sample_rate):
        o = oracle()
        o.ora_sync_create.restype = ctypes.c_void_p
        o.ora_sync_create.argtypes = [ctypes.c_float]
        o.ora_sync_destroy.argtypes = [ctypes.c_void_p]
        o.ora_sync_frequency.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int]
        o.ora_sync_symbol.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
        o.ora_sync_resample.restype = ctypes.c_double
        o.ora_sync_resample.argtypes = [ctypes.c_void_p]
        o.ora_sync_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.o = o
        self.h = o.ora_sync_create(sample_rate)

    def frequency(self, frequency_est, fft_size):
        self.o.ora_sync_frequency(self.h, frequency_est, fft_size)

    def symbol(self, phase_est, sample_rate_est):
        self.o.ora_sync_symbol(self.h, phase_est, sample_rate_est)

    def resample(self):
        return self.o.ora_sync_resample(self.h)

    def get(self):
        v = np.zeros(4, np.float64)
        self.o.ora_sync_get(self.h, v.ctypes.data)
        return dict(phase_est_filtered=np.float32(v[0]), frequency_est_filtered=np.float32(v[1]),
                    sample_rate_est_filtered=v[2], resample=v[3])

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_sync_destroy(self.h)
            self.h = None


class OraP1(object):
    """p1_symbol restated (oracle/p1_oracle.c)."""

    def __init__(self):
        o = oracle()
        o.ora_p1_create.restype = ctypes.c_void_p
        o.ora_p1_destroy.argtypes = [ctypes.c_void_p]
        o.ora_p1_execute.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.o = o
        self.h = o.ora_p1_create()
        self.buffer_sym = np.zeros(4096, np.complex64)

    def execute(self, x, consume=0, gain_changed=False, level_detect=0.0, reset=False, want_trace=False):
        x = _c64(x)
        c = ctypes.c_int(consume)
        res = (ctypes.c_int * 9)()
        coarse = ctypes.c_double()
        tr = np.full(len(x), np.nan, np.float32)
        fft = np.zeros(1024, np.complex64)
        d = self.o.ora_p1_execute(self.h, int(gain_changed), float(level_detect), len(x), x.ctypes.data, ctypes.byref(c),
                                  self.buffer_sym.ctypes.data, res, ctypes.byref(coarse), int(reset), tr.ctypes.data, fft.ctypes.data)
        out = dict(detected=bool(d), consume=c.value, idx_buffer_sym=res[0], p1_decoded=res[1], preamble=res[2], fft_mode=res[3],
                   s1=res[4], s2=res[5], shift=res[6], coarse_freq_offset=coarse.value)
        if want_trace:
            out["trace"], out["p1_fft"] = tr, fft
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.o.ora_p1_destroy(self.h)
            self.h = None


# ---- the reference's own Qt classes compiled where they lie (oracle/_ref/libref_t2sym.so, libref_t2rx.so; container only) ----
def _ref_lib(name):
    if name not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", "lib%s.so" % name)
        try:
            _cache[name] = ctypes.CDLL(path) if os.path.exists(path) else None
        except OSError:                                    # e.g. on the GPU box: the FFTW binary the reference ships is not there
            _cache[name] = None
    return _cache[name]


DVBT2_FIELDS = ("preamble", "bandwidth", "miso", "miso_group", "fft_mode", "fft_size", "guard_interval_mode", "guard_interval_size",
                "carrier_mode", "l_nulls", "pilot_pattern", "papr_mode", "l1_mod", "l1_cod", "l1_fec_type", "l1_post_size",
                "l1_post_info_size", "c_p2", "n_p2", "c_data", "c_fc", "n_fc", "k_total", "k_ext", "k_offset", "len_frame", "n_data",
                "n_t2", "l_fc", "t2_version")
L1_PRE_NAMES = ("type", "bwt_ext", "s1", "s2_field1", "s2_field2", "l1_repetition_flag", "guard_interval", "papr", "l1_post_mod", "l1_cod",
                "l1_fec_type", "l1_post_size", "l1_post_info_size", "pilot_pattern", "tx_id_availability", "cell_id", "network_id",
                "t2_system_id", "num_t2_frames", "num_data_symbols", "regen_flag", "l1_post_extension", "num_rf", "current_rf_index",
                "t2_version", "l1_post_scrambled", "t2_base_lite", "reserved", "crc_32")


class RefSym(object):
    """The reference's pilot_generator / address_freq_deinterleaver / p2_symbol / data_symbol / fc_symbol, driven in the order
    dvbt2_demodulator drives them (oracle/ref_t2sym.cpp). None-returning constructor helper: RefSym.open(...) gives None when the
    library is not available."""

    @staticmethod
    def available():
        return _ref_lib("ref_t2sym") is not None

    def __init__(self, preamble, fft_mode, strict=False):
        L = _ref_lib("ref_t2sym_strict" if strict else "ref_t2sym")
        L.ref_sym_new.restype = ctypes.c_void_p
        L.ref_sym_new.argtypes = [ctypes.c_int, ctypes.c_int]
        for fn, n in (("ref_sym_params", 2), ("ref_sym_data_init", 1)):
            getattr(L, fn).argtypes = [ctypes.c_void_p] * n
        L.ref_sym_set_l1_pre.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
        L.ref_sym_p2.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6
        L.ref_sym_data.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
        L.ref_sym_fc.argtypes = [ctypes.c_void_p] * 4
        L.ref_sym_carriers.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ref_sym_freq_deint.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.L, self.h = L, L.ref_sym_new(preamble, fft_mode)

    def params(self):
        v = np.zeros(30, np.int32)
        self.L.ref_sym_params(self.h, v.ctypes.data)
        return dict(zip(DVBT2_FIELDS, (int(x) for x in v)))

    def set_l1_pre(self, bwt_ext, guard_interval, papr, pilot_pattern, num_data_symbols):
        self.L.ref_sym_set_l1_pre(self.h, bwt_ext, guard_interval, papr, pilot_pattern, num_data_symbols)

    def data_init(self):
        return self.L.ref_sym_data_init(self.h)

    def p2(self, ofdm_cell, demod_init):
        p = self.params()
        x = np.ascontiguousarray(ofdm_cell, np.complex64)
        assert x.size == p["fft_size"]
        out = np.zeros(p["c_p2"], np.complex64)
        pre, post = np.zeros(29, np.int32), np.zeros(4096, np.int32)
        flags, sync = np.zeros(2, np.int32), np.zeros(2, np.float32)
        n = self.L.ref_sym_p2(self.h, int(demod_init), x.ctypes.data, out.ctypes.data, pre.ctypes.data, post.ctypes.data,
                              flags.ctypes.data, sync.ctypes.data)
        return dict(cells=out, l1_pre=dict(zip(L1_PRE_NAMES, (int(v) for v in pre))) if flags[0] else None,
                    l1_post=post[:n].copy() if n else None, crc_pre=bool(flags[0]), crc_post=bool(flags[1]),
                    sample_rate_offset=np.float32(sync[0]), phase_offset=np.float32(sync[1]))

    def data(self, idx_symbol, ofdm_cell):
        p = self.params()
        x = np.ascontiguousarray(ofdm_cell, np.complex64)
        out, sync = np.zeros(p["c_data"], np.complex64), np.zeros(2, np.float32)
        self.L.ref_sym_data(self.h, idx_symbol, x.ctypes.data, out.ctypes.data, sync.ctypes.data)
        return out, np.float32(sync[0]), np.float32(sync[1])

    def fc(self, ofdm_cell):
        p = self.params()
        x = np.ascontiguousarray(ofdm_cell, np.complex64)
        out, sync = np.zeros(p["n_fc"], np.complex64), np.zeros(2, np.float32)
        self.L.ref_sym_fc(self.h, x.ctypes.data, out.ctypes.data, sync.ctypes.data)
        return out, np.float32(sync[0]), np.float32(sync[1])

    def carriers(self, kind, idx_symbol):
        k = self.params()["k_total"]
        mp, rf = np.zeros(k, np.int32), np.zeros(k, np.float32)
        self.L.ref_sym_carriers(self.h, kind, idx_symbol, mp.ctypes.data, rf.ctypes.data)
        return mp, rf

    def freq_deint(self, kind):
        p = self.params()
        n = (p["c_p2"], p["c_data"], p["n_fc"])[kind]
        he, ho = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.L.ref_sym_freq_deint(self.h, kind, he.ctypes.data, ho.ctypes.data)
        return he, ho


def unpack_l1_post(v):
    """ref_sym_p2's flat l1_post -> dict (scalars, rf list, plp list with their dynamic part, aux list)."""
    names = ("sub_slices_per_frame", "num_plp", "num_aux", "aux_config_rfu", "fef_type", "fef_length", "fef_interval", "fef_length_msb",
             "reserved_2", "frame_idx", "sub_slice_interval", "type_2_start", "l1_change_counter", "start_rf_idx", "dyn_reserved_1",
             "dyn_reserved_3", "num_rf")
    v = [int(x) for x in v]
    d = dict(zip(names, v[:17]))
    p = 17
    d["rf"] = [(v[p + 2 * i], v[p + 2 * i + 1]) for i in range(d["num_rf"])]
    p += 2 * d["num_rf"]
    plp_names = ("id", "plp_type", "plp_payload_type", "ff_flag", "first_rf_idx", "first_frame_idx", "plp_group_id", "plp_cod", "plp_mod",
                 "plp_rotation", "plp_fec_type", "plp_num_blocks_max", "frame_interval", "time_il_length", "time_il_type",
                 "in_band_a_flag", "in_band_b_flag", "reserved_1", "plp_mode", "static_flag", "static_padding_flag",
                 "dyn_id", "dyn_start", "dyn_num_blocks", "dyn_reserved_2")
    d["plp"] = []
    for _ in range(d["num_plp"]):
        d["plp"].append(dict(zip(plp_names, v[p:p + 25])))
        p += 25
    d["aux"] = [(v[p + 2 * i], v[p + 2 * i + 1]) for i in range(d["num_aux"])]
    return d


def pack_l1_post(plps, frame_idx=0):
    """[(21 configurable ints as dict, dynamic dict)] -> the flat int array ref_t2rx's l1_holder reads."""
    names = ("id", "plp_type", "plp_payload_type", "ff_flag", "first_rf_idx", "first_frame_idx", "plp_group_id", "plp_cod", "plp_mod",
             "plp_rotation", "plp_fec_type", "plp_num_blocks_max", "frame_interval", "time_il_length", "time_il_type", "in_band_a_flag",
             "in_band_b_flag", "reserved_1", "plp_mode", "static_flag", "static_padding_flag")
    v = [len(plps)]
    for cfg, dyn in plps:
        v += [int(cfg.get(n, 0)) for n in names]
        v += [int(dyn.get("id", cfg.get("id", 0))), int(dyn.get("start", 0)), int(dyn.get("num_blocks", 0)), 0]
    v.append(frame_idx)
    return np.array(v, np.int32)


class _Taps(object):
    def _drain(self, fn, which, dtype):
        out = []
        meta = (ctypes.c_int * 4)()
        while True:
            n = fn(self.h, which, meta, None, 0)
            if n < 0:
                break
            buf = np.zeros(n, np.uint8)
            fn(self.h, which, meta, buf.ctypes.data, n)
            out.append((tuple(meta), buf.view(dtype).copy()))
        return out


class RefFec(_Taps):
    """The reference's time_deinterleaver -> llr_demapper -> ldpc_decoder -> bch_decoder -> bb_de_header chain on its own
    QThreads (oracle/ref_t2rx.cpp). taps: 0 ti_block cells, 1 LLR batches, 2 LDPC bits, 3 descrambled BBFRAMEs, 5 messages."""

    @staticmethod
    def available(strict=False):
        return _ref_lib("ref_t2rx_strict" if strict else "ref_t2rx") is not None

    def __init__(self, ts_path, need_plp=0, strict=False):
        L = _ref_lib("ref_t2rx_strict" if strict else "ref_t2rx")
        L.ref_fec_new.restype = ctypes.c_void_p
        L.ref_fec_new.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.ref_fec_keep.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.ref_fec_start.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.ref_fec_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.ref_fec_cells.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.ref_fec_tap.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ref_fec_close.argtypes = [ctypes.c_void_p]
        L.ref_fec_ldpc_execute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.L, self.ts_path = L, ts_path
        self.h = L.ref_fec_new(ts_path.encode(), need_plp)

    def start(self, l1_post_size, l1_post):
        self.L.ref_fec_start(self.h, l1_post_size, l1_post.ctypes.data)

    def frame(self, l1_post, cells):
        c = np.ascontiguousarray(cells, np.complex64)
        self.L.ref_fec_frame(self.h, l1_post.ctypes.data, c.size, c.ctypes.data)

    def cells(self, cells):
        c = np.ascontiguousarray(cells, np.complex64)
        self.L.ref_fec_cells(self.h, c.size, c.ctypes.data)

    def keep(self, which, on):
        self.L.ref_fec_keep(self.h, which, int(on))

    def ldpc_execute(self, l1_post, llr32, plp_simd=None):
        """ldpc_decoder::execute (the public slot, ldpc_decoder.h:90) on one SIMD batch: int8 [32][fec_size]."""
        a = np.ascontiguousarray(llr32, np.int8)
        assert a.shape[0] == 32
        idx = np.zeros(32, np.int32) if plp_simd is None else np.ascontiguousarray(plp_simd, np.int32)
        self.L.ref_fec_ldpc_execute(self.h, l1_post.ctypes.data, idx.ctypes.data, a.size, a.ctypes.data)

    def taps(self, which):
        dt = {0: np.complex64, 1: np.int8, 2: np.uint8, 3: np.uint8, 5: np.uint8}[which]
        return self._drain(self.L.ref_fec_tap, which, dt)

    def ts(self):
        self.L.ref_fec_close(self.h)
        return np.fromfile(self.ts_path, np.uint8)


class RefBbdh(_Taps):
    """bb_de_header alone (bb_de_header.cpp:84-448): descrambled BBFRAME bits in, what it writes to its TS file out."""

    def __init__(self, ts_path, need_plp=0, num_plp=1):
        L = _ref_lib("ref_t2rx")
        L.ref_bbdh_new.restype = ctypes.c_void_p
        L.ref_bbdh_new.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.ref_bbdh_execute.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.ref_bbdh_tap.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ref_bbdh_close.argtypes = [ctypes.c_void_p]
        self.L, self.ts_path = L, ts_path
        self.h = L.ref_bbdh_new(ts_path.encode(), need_plp, num_plp)

    def execute(self, plp_id, bits):
        b = np.ascontiguousarray(bits, np.uint8)
        self.L.ref_bbdh_execute(self.h, plp_id, b.size, b.ctypes.data)

    def messages(self):
        return [bytes(b).decode() for _, b in self._drain(self.L.ref_bbdh_tap, 5, np.uint8)]

    def ts(self):
        self.L.ref_bbdh_close(self.h)
        return np.fromfile(self.ts_path, np.uint8)


RX_STATE = ("c1", "c2", "level_detect", "phase_nco", "frequency_nco", "phase_est_filtered", "frequency_est_filtered",
            "sample_rate_est_filtered", "resample", "next_symbol_type", "idx_symbol", "symbol_size", "idx_buffer_sym", "est_chunk", "chunk",
            "p2_init", "demodulator_init", "deint_start", "crc32_l1_pre", "guard_interval_size", "fft_size", "dc_re", "dc_im")
SIG_FIELDS = ("change_frequency", "coarse_freq_offset", "frequency_changed", "change_gain", "gain_offset", "gain_changed",
              "correct_resample", "reset", "p1_reset")


class RefRx(_Taps):
    """The reference's dvbt2_demodulator (whole receiver) fed through its slot execute(len, i, q, signal_estimate*)."""

    @staticmethod
    def available(strict=False):
        return _ref_lib("ref_t2rx_strict" if strict else "ref_t2rx") is not None

    def __init__(self, ts_path, id_device=0, sample_rate=64.0e6 / 7.0, need_plp=0, strict=False, lib=None):
        """lib: another build of the same driver -- "ref_t2rx_gpufec" / "ref_t2rx_gpu": the reference's objects with the slot bodies of
        integration/*_gpu.cpp (libt2gpu.so behind the reference's own signal flow; oracle/Makefile, target `binding`)."""
        L = _ref_lib(lib or ("ref_t2rx_strict" if strict else "ref_t2rx"))
        if L is None:
            raise OSError("oracle/_ref/lib%s.so does not load here" % (lib or "ref_t2rx"))
        L.ref_rx_new.restype = ctypes.c_void_p
        L.ref_rx_new.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_char_p, ctypes.c_int]
        L.ref_rx_keep.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.ref_rx_execute.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.ref_rx_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ref_rx_set_loops.argtypes = [ctypes.c_void_p] + [ctypes.c_float] * 6
        L.ref_rx_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.ref_rx_tap.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ref_rx_close.argtypes = [ctypes.c_void_p]
        self.L, self.ts_path = L, ts_path
        self.h = L.ref_rx_new(id_device, sample_rate, ts_path.encode(), need_plp)
        self.sig = np.zeros(9, np.float64)

    def keep(self, which, on):
        self.L.ref_rx_keep(self.h, which, int(on))

    def execute(self, i16, q16):
        i16 = np.ascontiguousarray(i16, np.int16)
        q16 = np.ascontiguousarray(q16, np.int16)
        self.L.ref_rx_execute(self.h, i16.size, i16.ctypes.data, q16.ctypes.data, self.sig.ctypes.data)

    def state(self):
        v = np.zeros(23, np.float64)
        self.L.ref_rx_state(self.h, v.ctypes.data)
        return dict(zip(RX_STATE, v))

    def set_loops(self, c1=0.0, c2=1.0, phase_est_filtered=0.0, frequency_est_filtered=0.0, phase_nco=0.0, frequency_nco=0.0):
        self.L.ref_rx_set_loops(self.h, c1, c2, phase_est_filtered, frequency_est_filtered, phase_nco, frequency_nco)

    def buffer(self, which, n):
        out = np.zeros(n, np.complex64)
        self.L.ref_rx_buffer(self.h, which, n, out.ctypes.data)
        return out

    def taps(self, which):
        dt = {0: np.complex64, 1: np.int8, 2: np.uint8, 3: np.uint8, 4: np.complex64, 5: np.uint8}[which]
        return self._drain(self.L.ref_rx_tap, which, dt)

    def ts(self):
        self.L.ref_rx_close(self.h)
        return np.fromfile(self.ts_path, np.uint8)

    TRAJ = ("next_symbol_type", "idx_symbol", "chunk", "phase_est_filtered", "frequency_est_filtered", "sample_rate_est_filtered", "resample",
            "phase_nco", "frequency_nco", "old_sample_rate_est", "sample_rate_offset_hz", "frequency_offset_hz", "phase_integral", "phase_k_p",
            "frequency_integral", "frequency_k_p")

    def traj(self):
        """Per symbol that reached the tracking loops (dvbt2_demodulator.cpp:429-444), behind their update: [n][len(TRAJ)] float64."""
        self.L.ref_rx_traj.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        n = self.L.ref_rx_traj(self.h, None, 0)
        out = np.zeros((n, len(self.TRAJ)), np.float64)
        if n:
            self.L.ref_rx_traj(self.h, out.ctypes.data, n)
        return out

    def run_recording_tuned(self, base_i, base_q, buf, cfo_hz, tuner_step):
        """rx_sdrplay::start's loop over a recording behind an EMULATED TUNER (tests/ref_cases.py, RX_OFFSET_CASES): when the demodulator
        asks for a move of the local oscillator (change_frequency, coarse_freq_offset) the SDR thread's bookkeeping runs as in
        rx_sdrplay.cpp:158-176 and from that buffer on the recording is rotated by the move, rounded to a multiple of tuner_step.
        Returns (per-buffer states, moves = [(buffer index, requested Hz, tuner total Hz)])."""
        import t2_tx
        s = self.sig
        rf = [626.0e6]
        tuned = [0.0]
        moves = []

        def reset():
            s[7] = 0; s[1] = 0.0; s[0] = 1; s[6] = 0.0; s[4] = 0; s[3] = 1
            rf[0] = 626.0e6

        def set_rf(k):
            if not s[2]:
                s[2] = 1
            if s[0]:
                s[0] = 0; s[2] = 0
                s[6] = s[1] / rf[0]
                rf[0] += s[1]
                if s[1] != 0.0:
                    tuned[0] += tuner_step * np.rint(s[1] / tuner_step)
                    moves.append((k, float(s[1]), tuned[0]))
        reset()
        set_rf(0)
        log = []
        for k in range(len(base_i) // buf):
            if s[7]:
                reset(); set_rf(k)
                continue
            set_rf(k)
            s[5] = 1
            i16, q16 = t2_tx.rx_offset_rotate(base_i[k * buf:(k + 1) * buf], base_q[k * buf:(k + 1) * buf], k * buf, cfo_hz - tuned[0])
            self.execute(i16, q16)
            log.append(self.state())
        return log, moves

    def run_recording(self, i16, q16, buf):
        """rx_sdrplay::start's loop (rx_sdrplay.cpp:135-261) over a recording: reset(), then per buffer set_rf_frequency /
        set_gain bookkeeping and execute(). A recording cannot be re-tuned, so it must carry no carrier offset worth a re-tune
        (|coarse_freq_offset| < 10 Hz). Returns the per-buffer states."""
        s = self.sig
        rf = [626.0e6]

        def reset():
            s[7] = 0; s[1] = 0.0; s[0] = 1; s[6] = 0.0; s[4] = 0; s[3] = 1

        def set_rf():
            if not s[2]:
                s[2] = 1
            if s[0]:
                s[0] = 0; s[2] = 0
                s[6] = s[1] / rf[0]
                rf[0] += s[1]
        reset()
        set_rf()
        log = []
        for pos in range(0, len(i16) - buf + 1, buf):
            if s[7]:
                reset(); set_rf()
                continue
            set_rf()
            s[5] = 1
            self.execute(i16[pos:pos + buf], q16[pos:pos + buf])
            log.append(self.state())
        return log
