"""BBFRAME sequences for the de-framer (bb_de_header::execute, bb_de_header.cpp:84-448) -- shared by the golden generator (runs the
reference's class), tests/test_ref_pins.py (oracle/bbdh_oracle.c) and the product's host de-framer. Built from EN 302 755 5.1
(mode adaptation): high-efficiency mode (sync bytes removed, CRC-8 ^ 1 in the header) and normal mode (each sync byte replaced by
the CRC-8 of the preceding user packet). All integer work: regenerates identically anywhere.

Normal mode: the reference consumes one CRC byte per packet boundary WITHOUT taking its 8 bits off DFL (bb_de_header.cpp:
290-321), so it reads 8 bits per packet past the signalled data field. The frames here leave that many zero padding bits behind the
data field (DFL < K_bch - 80), which keeps the reference's reads inside the frame and its output well defined."""
import numpy as np

import t2_tx

K_BCH = 7032                     # 16200, r = 1/2


def _header(dfl, syncd, mode_hem, upl=0, sync=0, sis=1, isi=0):
    hdr = [1, 1, sis, 1, 0, 0, 0, 0] + t2_tx.bits_of(isi, 8)          # TS, SIS/MIS, CCM, no ISSY, no NPD, EXT 00 | ISI
    hdr += t2_tx.bits_of(upl, 16) + t2_tx.bits_of(dfl, 16) + t2_tx.bits_of(sync, 8) + t2_tx.bits_of(syncd, 16)
    hdr += t2_tx.bits_of(t2_tx.crc8_d5(hdr) ^ (1 if mode_hem else 0), 8)
    return hdr


def hem_frames(ts, n_frames, k_bch=K_BCH):
    return t2_tx.bbframes_hem(ts, k_bch, n_frames)[0]


def nm_frames(ts, n_frames, k_bch=K_BCH):
    """Normal-mode BBFRAMEs: user packets of 188 bytes = CRC-8 of the previous packet's 187 payload bytes + those 187 bytes."""
    ts = np.asarray(ts, np.uint8).reshape(-1, 188)
    ups = []
    prev = 0
    for p in ts:
        ups.append(np.concatenate([[prev], p[1:]]).astype(np.uint8))
        prev = t2_tx.crc8_d5(np.unpackbits(p[1:]))
    flow = np.unpackbits(np.concatenate(ups))
    room = k_bch - 80
    dfl = ((room - 8 * (room // 1504 + 2)) // 8) * 8                   # leave the reference's over-read inside the frame
    frames = np.zeros((n_frames, k_bch), np.uint8)
    pos = 0
    for f in range(n_frames):
        syncd = (-pos) % 1504
        frames[f, :80] = _header(dfl, syncd, False, upl=1504, sync=0x47)
        frames[f, 80:80 + dfl] = flow[pos:pos + dfl]
        pos += dfl
    return frames


def cases():
    """name -> (k_bch, [frame bits], [plp id per frame])."""
    ts = t2_tx.ts_packets(60, 77)
    hem = hem_frames(ts, 8)
    nm = nm_frames(ts, 8)
    out = {}
    out["hem_plain"] = (K_BCH, list(hem), [0] * 8)
    out["nm_plain"] = (K_BCH, list(nm), [0] * 8)
    for tag, fr in (("hem", hem), ("nm", nm)):
        other = fr.copy()                                                # frame 2 belongs to another PLP, frame 5 has a broken header
        broken = other[5].copy()
        broken[20] ^= 1
        plps = [0, 0, 1, 0, 0, 0, 0, 0]
        out[tag + "_other_plp_and_bad_crc"] = (K_BCH, [other[0], other[1], other[2], other[3], other[4], broken, other[6], other[7]], plps)
        out[tag + "_lost_frame_3"] = (K_BCH, [fr[i] for i in (0, 1, 2, 4, 5, 6)], [0] * 6)      # SYNCD disagrees with the pending packet
        out[tag + "_lost_frames_1_2"] = (K_BCH, [fr[i] for i in (0, 3, 4, 5)], [0] * 4)
        idle = fr[3].copy()                                              # SYNCD = 0xFFFF: no packet starts in this frame
        hdr = list(idle[:80])
        dfl = int("".join(str(b) for b in hdr[32:48]), 2)
        idle[:80] = _header(dfl, 0xFFFF, tag == "hem", upl=(0 if tag == "hem" else 1504), sync=(0 if tag == "hem" else 0x47))
        out[tag + "_syncd_ffff"] = (K_BCH, [fr[0], fr[1], fr[2], idle, fr[4], fr[5]], [0] * 6)
        # a data field that ends less than 8 bits into a packet: the frame leaves split = true with NO byte buffered; in normal
        # mode the reference then emits buffer[0] of an earlier frame regardless (bb_de_header.cpp:168-171)
        odd = fr.copy()
        hdr = list(odd[2, :80])
        syncd = int("".join(str(b) for b in hdr[56:72]), 2)
        dfl2 = syncd + (0 if tag == "hem" else 8) + 1504 * 2 + 4
        odd[2, :80] = _header(dfl2, syncd, tag == "hem", upl=(0 if tag == "hem" else 1504), sync=(0 if tag == "hem" else 0x47))
        out[tag + "_dfl_ends_under_a_byte"] = (K_BCH, [odd[i] for i in (0, 1, 2, 3, 4)], [0] * 5)
    mis = hem.copy()                                                     # multiple input streams: ISI carried in the header
    for f in range(4):
        hdr = _header(((K_BCH - 80) // 8) * 8, int("".join(str(b) for b in mis[f, 56:72]), 2), True, sis=0, isi=5)
        mis[f, :80] = hdr
    out["hem_mis"] = (K_BCH, list(mis[:4]), [0] * 4)
    return out
