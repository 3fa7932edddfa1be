/*
 * oracle/bbdh_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never linked into or called by the product library).
 * Plain-C restatement of the reference's BBFRAME de-framer as it is written, normal mode included with its accounting slip
 * (one CRC byte consumed per packet boundary without taking 8 bits off DFL, :290-321):
 *   bb_de_header::execute  /root/reference/src/DVB_T2/bb_de_header.cpp:84-448  (CRC-8 helpers :56-82, need_plp :139-142)
 * Pinned against the reference's own class (oracle/_ref/libref_t2rx.so built with the one-token define described in
 * oracle/Makefile; fixtures tests/golden/t2fec_golden.npz "bbdh/...": HEM and NM streams, lost frames, other PLP, broken header
 * CRC, SYNCD = 0xFFFF, multiple-input-stream header).
 * The reference reads past the frame when SYNCD / DFL lie; like the harness that drove the reference, the frame is copied in
 * front of 69 632 zero bytes so that those reads are defined. Its output buffer holds 53840/8 + 376 bytes; a frame that would
 * overrun it returns -4 here (undefined behaviour there).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TRANSPORT_PACKET_LENGTH 188
#define BIT_PACKET_LENGTH (TRANSPORT_PACKET_LENGTH * 8)
#define CRC_POLY 0xAB
#define CRC_POLYR 0xD5
#define TRANSPORT_ERROR_INDICATOR 0x80
#define OUT_LEN (53840 / 8 + TRANSPORT_PACKET_LENGTH * 2)

typedef struct {
    int need_plp, idx_packet, idx_buffer, split;
    uint8_t crc, crc_table[256], buffer[TRANSPORT_PACKET_LENGTH];
    int last_mode, resync, ts_error;
} ora_bbdh;

ora_bbdh *ora_bbdh_create(int need_plp)
{
    ora_bbdh *s = (ora_bbdh *)calloc(1, sizeof(ora_bbdh));
    s->need_plp = need_plp;
    for (int i = 0; i < 256; ++i) {                                          /* init_crc8_table (:56-68) */
        int r = i, crc = 0;
        for (int j = 7; j >= 0; --j) {
            if (((r & (1 << j)) ? 1 : 0) ^ ((crc & 0x80) ? 1 : 0)) crc = (crc << 1) ^ CRC_POLYR;
            else crc <<= 1;
        }
        s->crc_table[i] = (uint8_t)crc;
    }
    return s;
}
void ora_bbdh_destroy(ora_bbdh *s) { free(s); }
/* after a call: mode of the frame (0 NM, 1 HEM, -1 none), "resynchronizing" messages, "TS error." raised */
void ora_bbdh_info(const ora_bbdh *s, int *v3) { v3[0] = s->last_mode; v3[1] = s->resync; v3[2] = s->ts_error; }

static uint8_t take(const uint8_t **in)
{
    uint8_t t = 0;
    for (int n = 7; n >= 0; n--) t |= (uint8_t)(*(*in)++ << n);
    return t;
}

/* Returns the number of TS bytes written to out (the reference's datagram / file write of this call), -1 header CRC-8 error,
 * -2 frame skipped (other PLP or SYNCD 65535), -4 the reference's output buffer would overflow. */
int ora_bbdh_execute(ora_bbdh *s, int plp_id, int len_in, const uint8_t *bits, uint8_t *out_user, int out_cap)
{
    uint8_t *pad = (uint8_t *)calloc((size_t)len_in + 65536 + 4096, 1);
    memcpy(pad, bits, (size_t)len_in);
    const uint8_t *in = pad;
    uint8_t outbuf[OUT_LEN + 8192], *out = outbuf, *ptr_error_indicator = outbuf + OUT_LEN + 4096;   /* unset pointer: a scratch byte */
    int errors = 0, len_split = 0, len_out = 0, rc = 0, mode;
    s->resync = 0; s->ts_error = 0; s->last_mode = -1;
    {   /* check_crc8_mode over the 80 header bits (:70-82) */
        uint8_t crc = 0;
        for (int i = 0; i < 80; ++i) { uint8_t b = in[i] ^ (crc & 0x01); crc >>= 1; if (b) crc ^= CRC_POLY; }
        if (crc == 0) mode = 0; else if (crc == CRC_POLY) mode = 1; else { rc = -1; goto done; }
    }
    s->last_mode = mode;
    in += 16;                                                                /* MATYPE (:115-131): not used further */
    if (s->need_plp != plp_id) { rc = -2; goto done; }
    int upl = 0, dfl = 0, sync = 0, syncd = 0;
    for (int i = 15; i >= 0; --i) upl |= *in++ << i;
    for (int i = 15; i >= 0; --i) dfl |= *in++ << i;
    for (int i = 7; i >= 0; --i) sync |= *in++ << i;
    for (int i = 15; i >= 0; --i) syncd |= *in++ << i;
    (void)upl; (void)sync;
    if (syncd == 65535) { rc = -2; goto done; }
    in += 8;
#define PUT(v) do { if (len_out >= OUT_LEN) { rc = -4; goto done; } *out++ = (uint8_t)(v); ++len_out; } while (0)
    if (mode == 0) {                                                         /* INPUTMODE_NORMAL (:166-322) */
        if (s->split) {
            s->split = 0;
            PUT(s->buffer[0]);
            ptr_error_indicator = out;
            for (int i = 1; i < s->idx_buffer; ++i) PUT(s->buffer[i]);
            len_split = TRANSPORT_PACKET_LENGTH - s->idx_packet;
            int syncd_byte = syncd / 8;
            if (len_split == syncd_byte) {
                for (int i = 0; i < len_split; ++i) { uint8_t t = take(&in); s->crc = s->crc_table[t ^ s->crc]; PUT(t); ++s->idx_packet; }
                uint8_t t = take(&in);
                if (t != s->crc) { ++errors; *ptr_error_indicator |= TRANSPORT_ERROR_INDICATOR; }
                s->crc = 0;
            } else if (len_split < syncd_byte) {
                for (int i = 0; i < syncd_byte; ++i) { uint8_t t = take(&in); s->crc = s->crc_table[t ^ s->crc]; PUT(t); ++s->idx_packet; }
                uint8_t t = take(&in);
                if (t != s->crc) { ++errors; *ptr_error_indicator |= TRANSPORT_ERROR_INDICATOR; }
                s->crc = 0;
                ++s->resync;
            } else {
                for (int i = 0; i < syncd_byte; ++i) { uint8_t t = take(&in); PUT(t); ++s->idx_packet; }
                int dump = len_split - syncd_byte;
                for (int i = 0; i < dump; ++i) { PUT(0xF0); ++s->idx_packet; }
                ++errors;
                *ptr_error_indicator |= TRANSPORT_ERROR_INDICATOR;
                ++s->resync;
            }
        } else {
            in += syncd + 8;
        }
        dfl -= syncd + 8;
        while (dfl > 0) {
            if (dfl < BIT_PACKET_LENGTH) {
                s->split = 1;
                len_split = dfl / 8;
                s->idx_buffer = 0;
                for (int i = 0; i < len_split; ++i) {
                    if (s->idx_packet == TRANSPORT_PACKET_LENGTH) {
                        s->idx_packet = 0;
                        uint8_t t = take(&in);
                        if (t != s->crc) { ++errors; *ptr_error_indicator |= TRANSPORT_ERROR_INDICATOR; }
                        s->crc = 0;
                        s->buffer[s->idx_buffer++] = 0x47;
                        ++s->idx_packet;
                    }
                    uint8_t t = take(&in);
                    s->crc = s->crc_table[t ^ s->crc];
                    if (s->idx_buffer >= TRANSPORT_PACKET_LENGTH) { rc = -4; goto done; }
                    s->buffer[s->idx_buffer++] = t;
                    ++s->idx_packet;
                }
                dfl = 0;
            } else {
                if (s->idx_packet == TRANSPORT_PACKET_LENGTH) {
                    s->idx_packet = 0;
                    uint8_t t = take(&in);
                    if (t != s->crc) { ++errors; *ptr_error_indicator |= TRANSPORT_ERROR_INDICATOR; }
                    s->crc = 0;
                    PUT(0x47); ++s->idx_packet;
                    ptr_error_indicator = out;
                    t = take(&in); s->crc = s->crc_table[t ^ s->crc]; PUT(t); ++s->idx_packet;
                    dfl -= 8;
                } else if (s->idx_packet == 0) {
                    PUT(0x47); ++s->idx_packet;
                    ptr_error_indicator = out;
                    uint8_t t = take(&in); s->crc = s->crc_table[t ^ s->crc]; PUT(t); ++s->idx_packet;
                    dfl -= 8;
                } else {
                    uint8_t t = take(&in); s->crc = s->crc_table[t ^ s->crc]; PUT(t); ++s->idx_packet;
                    dfl -= 8;
                }
            }
        }
    } else {                                                                 /* INPUTMODE_HIEFF (:323-417) */
        if (s->split) {
            s->split = 0;
            for (int i = 0; i < s->idx_buffer; ++i) PUT(s->buffer[i]);
            len_split = TRANSPORT_PACKET_LENGTH - s->idx_packet;
            int syncd_byte = syncd / 8;
            if (len_split == syncd_byte) {
                for (int i = 0; i < len_split; ++i) { PUT(take(&in)); ++s->idx_packet; }
            } else if (len_split < syncd_byte) {
                for (int i = 0; i < len_split; ++i) { PUT(take(&in)); ++s->idx_packet; }
                in += syncd - len_split * 8;
                ++s->resync;
            } else {
                for (int i = 0; i < syncd_byte; ++i) { PUT(take(&in)); ++s->idx_packet; }
                int dump = len_split - syncd_byte;
                for (int i = 0; i < dump; ++i) { PUT(0xF0); ++s->idx_packet; }
                ++s->resync;
            }
        } else {
            in += syncd;
        }
        dfl -= syncd;
        while (dfl > 0) {
            if (dfl < BIT_PACKET_LENGTH) {
                s->split = 1;
                len_split = dfl / 8;
                s->idx_buffer = 0;
                for (int i = 0; i < len_split; ++i) {
                    if (s->idx_packet == TRANSPORT_PACKET_LENGTH) { s->idx_packet = 0; s->buffer[s->idx_buffer++] = 0x47; ++s->idx_packet; }
                    if (s->idx_buffer >= TRANSPORT_PACKET_LENGTH) { rc = -4; goto done; }
                    s->buffer[s->idx_buffer++] = take(&in);
                    ++s->idx_packet;
                }
                dfl = 0;
            } else {
                if (s->idx_packet == TRANSPORT_PACKET_LENGTH || s->idx_packet == 0) {
                    s->idx_packet = 0;
                    PUT(0x47); ++s->idx_packet;
                } else {
                    PUT(take(&in)); ++s->idx_packet;
                    dfl -= 8;
                }
            }
        }
    }
#undef PUT
    if (errors != 0) s->ts_error = 1;
    if (len_out > out_cap) { rc = -4; goto done; }
    memcpy(out_user, outbuf, (size_t)len_out);
    rc = len_out;
done:
    free(pad);
    return rc;
}
