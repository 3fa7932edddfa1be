// bb_deheader.cpp -- host-side BBFRAME de-framing into transport-stream bytes (C-ABI t2gpu_bbdh_*).
//
// Replaces bb_de_header::execute (/root/reference/src/DVB_T2/bb_de_header.cpp:84-448; check_crc8_mode :70-82,
// init_crc8_table :56-68). This stage is sequential by nature (a TS packet may straddle two BBFRAMEs, the running CRC-8 of
// normal mode crosses the boundary too) and costs a few kilobytes per frame: it stays on the host exactly as in the
// reference, consuming the descrambled bits the GPU stages deliver. The reference sends the bytes to a UDP socket or a
// file (:433-443); here they are returned to the caller.
#include "../../include/t2gpu.h"
#include "t2gpu_common.h"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace t2gpu;

namespace {
const int TS_LEN = 188, BIT_PACKET = 188 * 8, BBH_BITS = 80;
const uint8_t CRC_POLY = 0xAB, TEI = 0x80;
}

struct t2gpu_bbdh {
    int need_plp = 0;
    uint8_t crc_table[256];
    uint8_t hdr_table[256];                            // check_crc8_mode (:70-82) a byte at a time: state' = hdr_table[state ^ bit-reversed byte]
    uint8_t rev8[256];
    uint8_t crc = 0;
    int idx_packet = 0, idx_buffer = 0;
    bool split = false;
    uint8_t buffer[188];
    int last_mode = -1;
    int resync = 0;                                    // "Baseband header resynchronizing." raised by the last call (:218,235,369,384)
    std::vector<uint8_t> packed;                       // t2gpu_bbdh_execute: the frame with eight bits per byte
};

// Bit reader bounded by the frame: the reference trusts SYNCD / DFL and, in normal mode, consumes 8 bits per packet more than it
// takes off DFL (:290-321), so it reads past the frame it was given. Bits past the end read as 0 here (what the reference sees
// when the frame sits in a zeroed buffer) -- never memory outside the frame. Two input forms: one bit per byte (the reference's
// stage interface, bch_decoder.h:41 -> bb_de_header.h:59) and packed, MSB first (what K-descramble-pack delivers: k_bch / 8 bytes
// per BBFRAME instead of k_bch, and a byte-aligned run is one load per output byte).
struct BitSrc {
    const uint8_t *base;
    long pos, len;                                     // in bits
    bool packed;
    int bit(long i) const
    {
        if (i < 0 || i >= len) return 0;
        return packed ? (base[i >> 3] >> (7 - (i & 7))) & 1 : base[i] & 1;
    }
    uint8_t byte()
    {
        uint8_t t;
        if (packed && (pos & 7) == 0 && pos >= 0 && pos + 8 <= len) t = base[pos >> 3];
        else {
            t = 0;
            for (int n = 7; n >= 0; --n) t |= (uint8_t)(bit(pos + (7 - n)) << n);
        }
        pos += 8;
        return t;
    }
    void skip(long n) { pos += n; }                    // may run past the end: byte() then yields zeros
    // n whole bytes that lie byte-aligned inside a packed frame: the caller may copy them in one piece
    const uint8_t *run(int n) const { return (packed && (pos & 7) == 0 && pos >= 0 && pos + 8L * n <= len) ? base + (pos >> 3) : nullptr; }
};
// Byte sink bounded by out_cap: the reference's buffer is a fixed 53840/8 + 376 bytes (:35-38) which a wild SYNCD overruns; here
// the frame is refused instead (full = true -> -3).
struct ByteSink {
    uint8_t *o, *end;
    int n = 0;
    bool full = false;
    void put(uint8_t v) { if (o < end) { *o++ = v; ++n; } else full = true; }
};

extern "C" t2gpu_bbdh *t2gpu_bbdh_create(int need_plp)
{
    t2gpu_bbdh *h = new t2gpu_bbdh();
    h->need_plp = need_plp;
    for (int i = 0; i < 256; ++i) {                    // CRC-8, generator 0xD5, MSB first (bb_de_header.cpp:56-68)
        int r = i, crc = 0;
        for (int j = 7; j >= 0; --j) {
            if (((r >> j) & 1) ^ ((crc & 0x80) ? 1 : 0)) crc = (crc << 1) ^ 0xD5;
            else crc <<= 1;
        }
        h->crc_table[i] = (uint8_t)crc;
        // the header check shifts right and takes the frame's bits in order, i.e. a packed byte's MSB first: eight steps of it from
        // state c on byte v give T[c ^ rev8(v)], T = eight steps from state x on zero input (the update is linear in (state, input))
        uint8_t c = (uint8_t)i;
        for (int k = 0; k < 8; ++k) { const uint8_t b = c & 1; c >>= 1; if (b) c ^= CRC_POLY; }
        h->hdr_table[i] = c;
        uint8_t rv = 0;
        for (int k = 0; k < 8; ++k) rv |= (uint8_t)(((i >> k) & 1) << (7 - k));
        h->rev8[i] = rv;
    }
    return h;
}
extern "C" void t2gpu_bbdh_destroy(t2gpu_bbdh *h) { delete h; }
extern "C" int t2gpu_bbdh_mode(const t2gpu_bbdh *h) { return h ? h->last_mode : -1; }

namespace {
int bbdh_run(t2gpu_bbdh *h, int plp_id, int len_in, const uint8_t *bits, bool packed, uint8_t *out, int out_cap, int *ts_errors)
{
    if (!h || !bits || !out || len_in < BBH_BITS || out_cap < len_in / 8 + 2 * TS_LEN) { set_error("t2gpu_bbdh_execute: bad arguments"); return -3; }
    h->resync = 0;                                     // status of THIS call, whichever way it returns
    if (ts_errors) *ts_errors = 0;
    BitSrc in{bits, 0, len_in, packed};
    ByteSink snk{out, out + out_cap};
    int errors = 0;
    uint8_t *tei = nullptr;
    // BBHEADER CRC over its 80 bits: remainder 0 = normal mode, 0xAB = high-efficiency mode (CRC-8 xor MODE), :70-82,101-113
    uint8_t c = 0;
    if (packed) {
        for (int i = 0; i < BBH_BITS / 8; ++i) c = h->hdr_table[c ^ h->rev8[bits[i]]];
    } else {
        for (int i = 0; i < BBH_BITS; ++i) {
            uint8_t b = (uint8_t)(in.bit(i) ^ (c & 0x01));
            c >>= 1;
            if (b) c ^= CRC_POLY;
        }
    }
    int hem;
    if (c == 0) hem = 0;
    else if (c == CRC_POLY) hem = 1;
    else return -1;                                    // "Baseband header CRC8 error.": frame dropped
    h->last_mode = hem;
    if (h->need_plp != plp_id) return -2;              // :139-142
    long hp = 16;                                      // MATYPE (TS/GS, SIS/MIS, CCM/ACM, ISSYI, NPD, EXT | ISI): not used by the data path
    int upl = 0, dfl = 0, sync = 0, syncd = 0;
    if (packed) {                                      // the same fields from whole bytes (MSB-first packing = the fields' own bit order)
        upl = bits[2] << 8 | bits[3]; dfl = bits[4] << 8 | bits[5]; sync = bits[6]; syncd = bits[7] << 8 | bits[8];
    } else {
        for (int i = 15; i >= 0; --i) upl |= in.bit(hp++) << i;
        for (int i = 15; i >= 0; --i) dfl |= in.bit(hp++) << i;
        for (int i = 7; i >= 0; --i) sync |= in.bit(hp++) << i;
        for (int i = 15; i >= 0; --i) syncd |= in.bit(hp++) << i;
    }
    (void)upl; (void)sync;
    if (syncd == 65535) return -2;                     // no user packet starts in this frame (:160-163)
    in.skip(BBH_BITS);

    if (!hem) {                                        // ---- normal mode (:166-322): CRC-8 of the previous packet replaces the sync byte
        if (h->split) {
            h->split = false;
            // as written (:168-171): buffer[0] goes out and the TEI pointer sits behind it whether or not the previous frame left
            // any byte (a DFL remainder under 8 bits leaves idx_buffer = 0: the byte is then whatever buffer[0] last held)
            snk.put(h->buffer[0]); tei = snk.full ? nullptr : snk.o;
            for (int i = 1; i < h->idx_buffer; ++i) snk.put(h->buffer[i]);
            const int len_split = TS_LEN - h->idx_packet, syncd_byte = syncd / 8;
            if (len_split <= syncd_byte) {
                const int take = (len_split == syncd_byte) ? len_split : syncd_byte;     // as written: the longer run when they disagree
                for (int i = 0; i < take; ++i) {
                    uint8_t t = in.byte();
                    h->crc = h->crc_table[t ^ h->crc];
                    snk.put(t); ++h->idx_packet;
                }
                uint8_t t = in.byte();
                if (t != h->crc) { ++errors; if (tei && tei < snk.end) *tei |= TEI; }
                h->crc = 0;
                if (len_split != syncd_byte) ++h->resync;
            } else {
                for (int i = 0; i < syncd_byte; ++i) { snk.put(in.byte()); ++h->idx_packet; }
                for (int i = 0; i < len_split - syncd_byte; ++i) { snk.put(0xF0); ++h->idx_packet; }
                ++errors;
                if (tei && tei < snk.end) *tei |= TEI;
                ++h->resync;
            }
        } else {
            in.skip(syncd + 8);
        }
        dfl -= syncd + 8;
        while (dfl > 0 && !snk.full) {
            if (dfl < BIT_PACKET) {
                h->split = true;
                const int len_split = dfl / 8;
                h->idx_buffer = 0;
                for (int i = 0; i < len_split && h->idx_buffer < TS_LEN; ++i) {
                    if (h->idx_packet == TS_LEN) {
                        h->idx_packet = 0;
                        uint8_t t = in.byte();
                        if (t != h->crc) { ++errors; if (tei && tei < snk.end) *tei |= TEI; }
                        h->crc = 0;
                        h->buffer[h->idx_buffer++] = 0x47;
                        ++h->idx_packet;
                        if (h->idx_buffer == TS_LEN) break;
                    }
                    uint8_t t = in.byte();
                    h->crc = h->crc_table[t ^ h->crc];
                    h->buffer[h->idx_buffer++] = t;
                    ++h->idx_packet;
                }
                dfl = 0;
            } else {
                if (h->idx_packet == TS_LEN) {
                    h->idx_packet = 0;
                    uint8_t t = in.byte();             // the CRC byte: consumed, but DFL is NOT reduced for it (the reference's slip, :290-300)
                    if (t != h->crc) { ++errors; if (tei && tei < snk.end) *tei |= TEI; }
                    h->crc = 0;
                }
                if (h->idx_packet == 0) {
                    snk.put(0x47); ++h->idx_packet;
                    tei = snk.full ? nullptr : snk.o;
                }
                uint8_t t = in.byte();
                h->crc = h->crc_table[t ^ h->crc];
                snk.put(t); ++h->idx_packet;
                dfl -= 8;
            }
        }
    } else {                                           // ---- high-efficiency mode (:323-417): 187-byte packets, sync re-inserted
        // n frame bytes to dst, in one piece where the frame is packed and the run byte-aligned inside it, else bit by bit as written
        auto take = [&in](uint8_t *dst, int n) {
            const uint8_t *src = n > 8 ? in.run(n) : nullptr;
            if (src) { std::memcpy(dst, src, (size_t)n); in.skip(8L * n); }
            else for (int i = 0; i < n; ++i) dst[i] = in.byte();
        };
        auto emit = [&](int n) {                      // ... to the output (the reference's `*out++ = temp` loops)
            if (n > 0 && snk.end - snk.o >= n) { take(snk.o, n); snk.o += n; snk.n += n; }
            else for (int i = 0; i < n; ++i) snk.put(in.byte());
        };
        if (h->split) {
            h->split = false;
            if (snk.end - snk.o >= h->idx_buffer) { std::memcpy(snk.o, h->buffer, (size_t)h->idx_buffer); snk.o += h->idx_buffer; snk.n += h->idx_buffer; }
            else for (int i = 0; i < h->idx_buffer; ++i) snk.put(h->buffer[i]);
            const int len_split = TS_LEN - h->idx_packet, syncd_byte = syncd / 8;
            if (len_split <= syncd_byte) {
                emit(len_split); h->idx_packet += len_split;
                if (len_split < syncd_byte) { in.skip(syncd - len_split * 8); ++h->resync; }
            } else {
                emit(syncd_byte); h->idx_packet += syncd_byte;
                for (int i = 0; i < len_split - syncd_byte; ++i) { snk.put(0xF0); ++h->idx_packet; }
                ++h->resync;
            }
        } else {
            in.skip(syncd);
        }
        dfl -= syncd;
        while (dfl > 0 && !snk.full) {
            if (dfl < BIT_PACKET) {
                h->split = true;
                const int len_split = dfl / 8;
                h->idx_buffer = 0;
                for (int i = 0; i < len_split && h->idx_buffer < TS_LEN;) {      // the reference's byte loop (:389-403), run by run
                    if (h->idx_packet == TS_LEN) {
                        h->idx_packet = 0;
                        h->buffer[h->idx_buffer++] = 0x47;
                        ++h->idx_packet;
                        if (h->idx_buffer == TS_LEN) break;
                    }
                    const int n = std::min(std::min(len_split - i, TS_LEN - h->idx_packet), TS_LEN - h->idx_buffer);
                    take(h->buffer + h->idx_buffer, n);
                    h->idx_buffer += n; h->idx_packet += n; i += n;
                }
                dfl = 0;
            } else {
                if (h->idx_packet == TS_LEN || h->idx_packet == 0) {
                    h->idx_packet = 0;
                    snk.put(0x47); ++h->idx_packet;
                } else {
                    // the bytes up to the end of this packet that the loop would emit one by one (each needs dfl >= BIT_PACKET before
                    // it): from a packed, byte-aligned frame they move in one piece
                    const int run = std::min(TS_LEN - h->idx_packet, (dfl - BIT_PACKET) / 8 + 1);
                    const uint8_t *src = run > 8 ? in.run(run) : nullptr;
                    if (src && snk.end - snk.o >= run) {
                        std::memcpy(snk.o, src, (size_t)run);
                        snk.o += run; snk.n += run; in.skip(8L * run);
                        h->idx_packet += run; dfl -= 8 * run;
                    } else {
                        snk.put(in.byte()); ++h->idx_packet;
                        dfl -= 8;
                    }
                }
            }
        }
    }
    if (snk.full) {                                    // SYNCD / DFL ask for more bytes than the contract's out_cap: refuse, resynchronise on the next frame
        h->split = false; h->idx_packet = 0; h->idx_buffer = 0; h->crc = 0;
        set_error("t2gpu_bbdh_execute: SYNCD/DFL of this frame would overrun the output buffer");
        return -3;
    }
    if (ts_errors) *ts_errors = errors;
    return snk.n;
}
}  // namespace

// One bit per byte in (the reference's stage interface): the frame is packed first -- eight bytes to one with a multiplication -- and
// goes through the packed form, whose byte-aligned runs are one load per output byte (the bit-by-bit reader cost 28 us per 64800-bit
// frame, a third of the slot-shaped path's time once the transport stream came out). Same bytes: both forms read bit i as bits[i] & 1
// and zeros behind the frame.
extern "C" int t2gpu_bbdh_execute(t2gpu_bbdh *h, int plp_id, int len_in, const uint8_t *bits, uint8_t *out, int out_cap,
                                  int *ts_errors)
{
    if (!h || !bits || len_in < BBH_BITS) return bbdh_run(h, plp_id, len_in, bits, false, out, out_cap, ts_errors);   // its own argument check answers
    const size_t nbytes = ((size_t)len_in + 7) / 8;
    if (h->packed.size() < nbytes) h->packed.resize(nbytes);
    uint8_t *pk = h->packed.data();
    const int whole = len_in / 8;
    for (int i = 0; i < whole; ++i) {
        uint64_t x;
        std::memcpy(&x, bits + 8 * (size_t)i, 8);
        pk[i] = (uint8_t)(((x & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);    // byte k (bit 8k) lands on bit 63 - k
    }
    if (len_in & 7) {
        uint8_t t = 0;
        for (int k = 0; k < (len_in & 7); ++k) t |= (uint8_t)((bits[8 * (size_t)whole + k] & 1) << (7 - k));
        pk[whole] = t;
    }
    return bbdh_run(h, plp_id, len_in, pk, true, out, out_cap, ts_errors);
}

// The same on a packed BBFRAME: len_in bits in (len_in + 7) / 8 bytes, MSB first (t2gpu_bch_descramble_pack_dev, t2gpu_rx).
extern "C" int t2gpu_bbdh_execute_packed(t2gpu_bbdh *h, int plp_id, int len_in, const uint8_t *bytes, uint8_t *out, int out_cap,
                                         int *ts_errors)
{
    return bbdh_run(h, plp_id, len_in, bytes, true, out, out_cap, ts_errors);
}

// A whole run of packed BBFRAMEs in stream order through the ONE de-framer (the rank-0 end of a frame-sharded receiver, and any
// caller that has many rows at hand): rows[i] = k_bch bits in k_bch / 8 bytes at rows + i * row_stride; trials (may be null) holds the
// LDPC verdict of every SIMD batch of `group` rows -- the rows of a batch the decoder gave up on never reach bb_de_header in the
// reference (ldpc_decoder.cpp:264-268) and are skipped here. TS bytes are appended to out; counts (may be null) receives
// {rows de-framed, rows dropped by the LDPC rule, BBHEADER CRC errors, frames skipped (other PLP / SYNCD 65535), TS packet errors,
// resynchronisations}. Returns the number of TS bytes, or -3 (bad arguments / out_cap too small for what the rows carry).
extern "C" long t2gpu_bbdh_execute_packed_rows(t2gpu_bbdh *h, int plp_id, int k_bch, const uint8_t *rows, long n_rows, long row_stride,
                                               const int32_t *trials, int group, uint8_t *out, long out_cap, long *counts)
{
    if (!h || !rows || !out || n_rows < 0 || k_bch < BBH_BITS || row_stride < (k_bch + 7) / 8 || (trials && group < 1)) {
        set_error("t2gpu_bbdh_execute_packed_rows: bad arguments");
        return -3;
    }
    const long per = k_bch / 8 + 2 * TS_LEN;                   // the most one frame can emit, and bbdh_run's own out_cap contract
    long used = 0, c[6] = {0, 0, 0, 0, 0, 0};
    for (long i = 0; i < n_rows; ++i) {
        if (trials && trials[i / group] < 0) { ++c[1]; continue; }
        if (out_cap - used < per) { set_error("t2gpu_bbdh_execute_packed_rows: out_cap too small"); return -3; }
        int err = 0;
        const int n = bbdh_run(h, plp_id, k_bch, rows + i * row_stride, true, out + used, (int)per, &err);
        ++c[0];
        c[5] += h->resync;
        if (n > 0) { used += n; c[4] += err; }
        else if (n == -1) ++c[2];
        else if (n == -2 || n == -3) ++c[3];            // -3 here: a frame whose SYNCD / DFL would overrun its share of out -- refused, state reset
    }
    if (counts) for (int k = 0; k < 6; ++k) counts[k] = c[k];
    return used;
}

extern "C" int t2gpu_bbdh_resync_count(const t2gpu_bbdh *h) { return h ? h->resync : 0; }
extern "C" int t2gpu_bbdh_reset(t2gpu_bbdh *h)
{
    if (!h) return -3;
    h->crc = 0; h->idx_packet = 0; h->idx_buffer = 0; h->split = false; h->last_mode = -1; h->resync = 0;
    std::memset(h->buffer, 0, sizeof h->buffer);
    return 0;
}
