/*
 * oracle/ddn_oracle_ysf.c - TEST INFRASTRUCTURE ONLY: CPU restatement of the Yaesu System Fusion frame information channel decode,
 * the second consumer of SURVEY 8a row a17 (the libM17-style K = 5 decoder).
 *
 *   ysf_conv_fich()   src/protocol/ysf/ysf.c:357-424: the 100 FICH dibits behind a frame sync -> dibit de-interleave (20 x 5) ->
 *                     dsd_ysf_soft_viterbi_decode(buf, 100, 13, 8, 96) = hard costs 0 / 65535 per bit through
 *                     viterbi_decode_punctured() with a pattern of all ones (src/protocol/ysf/ysf_frame.c:74-126), bits 8 .. 103 of
 *                     the 14 decoded bytes -> four Golay(24,12) words (Golay_24_12_decode, src/fec/fec.c:656-690) -> 48 bits ->
 *                     ysf_crc16() :229-242 (poly 0x1021, bits shifted in, complemented; a good frame gives 0) -> the 32 FICH bits
 *   fields            ysf_parse_fich() :512-556: FI 2, CS 2, CM 2, BN 2, BT 2, FN 3, FT 3, rsv, MR 3, VoIP path 1, DT 2, SQL type 1,
 *                     SQL code 7; what DECODE_IQ_YSF asserts ("V/D2 RID Mode Repeater CC", tests/CMakeLists.txt:8953-8957) is
 *                     DT = 2, CM = 1, path = 1, FI = 1 (ysf_print_fich_* :562-619)
 *
 * PARITY STATUS: ysf_frame.c compiles from the reference's sources into oracle/_ref; the Viterbi stage here is checked against its
 * dsd_ysf_soft_viterbi_decode() and the Golay stage against the compiled Golay_24_12_decode() (tests/test_oracle_ysf.py).  ysf.c itself
 * needs the vocoder and the engine: ysf_conv_fich()'s body and ysf_crc16() are restated.
 */
#include <string.h>

#include "ddn_oracle.h"
#include "ddn_tables_ambe.h"

uint16_t
orc_ysf_crc16(const uint8_t* bits, int len) {
    const uint32_t poly = 0x1021;
    uint32_t crc = 0;
    for (int i = 0; i < len; i++) {
        const uint32_t bit = bits[i] & 1u;
        crc = ((crc << 1) | bit) & 0x1ffff;
        if (crc & 0x10000) {
            crc = (crc & 0xffff) ^ poly;
        }
    }
    crc = crc ^ 0xffff;
    return (uint16_t)(crc & 0xffff);
}

/* dsd_ysf_soft_viterbi_decode(dibits, n, decoded_bytes, offset_bits, output_bits): -> out_bits[output_bits]; returns the path cost */
uint32_t
orc_ysf_soft_viterbi(const uint8_t* dibits, int n, int decoded_bytes, int offset_bits, int output_bits, uint8_t* out_bits) {
    static const uint8_t none[4] = {1, 1, 1, 1};
    uint16_t soft[244 * 2];
    uint8_t dec[64];
    memset(soft, 0, sizeof(soft));
    memset(dec, 0, sizeof(dec));
    for (int i = 0; i < n; i++) {
        soft[2 * i] = (dibits[i] & 2) ? 0xFFFF : 0;
        soft[2 * i + 1] = (dibits[i] & 1) ? 0xFFFF : 0;
    }
    const uint32_t err = orc_m17_viterbi_decode_punctured(dec, soft, none, 2 * n, 4);
    for (int i = 0; i < output_bits; i++) {
        const int b = offset_bits + i;
        out_bits[i] = (b < 8 * (decoded_bytes + 1)) ? (uint8_t)((dec[b >> 3] >> (7 - (b & 7))) & 1) : 0;
    }
    return err;
}

/* ysf_conv_fich(): returns 0, -1 (a Golay word failed) or -2 (CRC); fich32 = the 32 information bits either way */
int
orc_ysf_fich(const uint8_t in100[100], uint8_t fich32[32], uint32_t* v_error) {
    uint8_t buf[100], tb[100], fich[48];
    for (int i = 0; i < 20; i++) {
        for (int j = 0; j < 5; j++) {
            buf[j + i * 5] = in100[i + j * 20];
        }
    }
    memset(tb, 0, sizeof(tb));
    const uint32_t ve = orc_ysf_soft_viterbi(buf, 100, 13, 8, 96, tb);
    if (v_error) {
        *v_error = ve;
    }
    int err = 0;
    for (int i = 0; i < 4; i++) {
        uint8_t w[24];
        memcpy(w, tb + 24 * i, 24);
        if (!orc_golay_dmr_decode(24, w)) {
            err = -1;
        }
        memcpy(tb + 24 * i, w, 24);
    }
    for (int i = 0; i < 12; i++) {
        fich[i] = tb[i];
        fich[12 + i] = tb[i + 24];
        fich[24 + i] = tb[i + 48];
        fich[36 + i] = tb[i + 72];
    }
    if (orc_ysf_crc16(fich, 48) != 0) {
        err = -2;
    }
    memcpy(fich32, fich, 32);
    return err;
}

/* ---- the frame's payload behind the FICH (round 5): V/D mode 2 voice channel bits, the data channels, the dispatch ----------------
 *   dsd_ysf_vd2_interleave_index / dsd_ysf_pn95_bit / dsd_ysf_dewhiten_bits   src/protocol/ysf/ysf_frame.c:33-73 (26 x 4 bit matrix;
 *                     PN9 x^9 + x^4 + 1 seeded 0x1C9, LSB out, restarted every 512 bits) - checked against the compiled file and the
 *                     numbers tests/protocol/ysf/test_ysf_frame.c asserts (tests/test_oracle_ysf.py)
 *   ysf_read_type2_vech_bits / ysf_build_type2_ambe   ysf.c:687-722: 52 dibits -> 104 de-interleaved, de-whitened bits -> 27 majority
 *                     votes over bit triples + 22 plain bits = the 49 AMBE parameter bits; errs2 = bit 103
 *   ysf_conv_dch2 / ysf_conv_dch   ysf.c:245-355: dibit de-interleave 20 x 5 / 20 x 9 -> dsd_ysf_soft_viterbi_decode(100, 13, 8, 96) /
 *                     (180, 23, 8, 176) -> ysf_crc16 over all decoded bits (0 = good) -> de-whiten the 80 / 160 data bits -> 10 / 20 bytes
 *   ysf_parse_fich's fall-back + ysf_dispatch_payload   ysf.c:512-556, 908-922: a frame whose FICH failed is read as the last good
 *                     frame's type; FI = 1: DT 0 -> V/D1 (5 x 36 data + 36 voice dibits), DT 2 -> V/D2 (5 x 20 data + 52 voice),
 *                     DT 3 -> full-rate voice; DT = 1 or FI = 0 / 2 -> full-rate data (10 x 36 dibits, two DCH blocks in turn)
 * PARITY STATUS of this part: the primitives are pinned to the compiled ysf_frame.c; ysf.c's callers are restated (ysf.c needs the
 * engine and the vocoder).  The reference's capture is V/D2: its DCH2 blocks pass CRC16 through this path (tests/test_oracle_ysf.py). */
int
orc_ysf_vd2_index(int k) {
    return (k % 4) * 26 + k / 4;
}

static int
pn95_next(unsigned* lfsr) {
    const int bit = (int)(*lfsr & 1u);
    const unsigned fb = ((*lfsr >> 4) ^ *lfsr) & 1u;
    *lfsr = (*lfsr >> 1) | (fb << 8);
    return bit;
}

int
orc_ysf_pn95_bit(int i) {
    unsigned l = 0x1C9;
    i %= 512;
    for (int k = 0; k < i; k++) {
        (void)pn95_next(&l);
    }
    return pn95_next(&l);
}

void
orc_ysf_dewhiten(uint8_t* bits, int n) {
    int off = 0;
    while (off < n) {
        unsigned l = 0x1C9;
        for (int i = 0; i < 512 && off < n; i++, off++) {
            bits[off] = (uint8_t)(bits[off] ^ pn95_next(&l));
        }
    }
}

/* 52 voice-channel dibits of one V/D2 sub-frame -> ambe_d[49]; returns errs2 (= de-whitened bit 103) */
int
orc_ysf_vd2_voice(const uint8_t dibits52[52], uint8_t ambe_d[49]) {
    static const uint8_t majority[8] = {0, 0, 0, 1, 0, 1, 1, 1};
    uint8_t v[104];
    int k = 0;
    for (int j = 0; j < 52; j++) {
        const int msb = orc_ysf_vd2_index(k++), lsb = orc_ysf_vd2_index(k++);
        v[msb] = (uint8_t)(((dibits52[j] >> 1) & 1) ^ orc_ysf_pn95_bit(msb));
        v[lsb] = (uint8_t)((dibits52[j] & 1) ^ orc_ysf_pn95_bit(lsb));
    }
    for (int t = 0; t < 27; t++) {
        ambe_d[t] = majority[(v[3 * t] << 2) | (v[3 * t + 1] << 1) | v[3 * t + 2]];
    }
    for (int j = 0; j < 22; j++) {
        ambe_d[27 + j] = v[81 + j];
    }
    return v[103];
}

/* ysf_conv_dch2 (n = 100) / ysf_conv_dch (n = 180): out = 10 / 20 de-whitened bytes; returns 1 good, 3 CRC16 failed */
int
orc_ysf_dch(const uint8_t* in, int n, uint8_t* out_bytes, uint32_t* v_error) {
    uint8_t buf[180], tb[192];
    const int cols = n / 20, nbits = n == 100 ? 96 : 176, nbytes = n == 100 ? 13 : 23;
    for (int i = 0; i < 20; i++) {
        for (int j = 0; j < cols; j++) {
            buf[j + i * cols] = in[i + j * 20];
        }
    }
    memset(tb, 0, sizeof(tb));
    const uint32_t ve = orc_ysf_soft_viterbi(buf, n, nbytes, 8, nbits, tb);
    if (v_error) {
        *v_error = ve;
    }
    const int st = orc_ysf_crc16(tb, nbits) == 0 ? 1 : 3;
    orc_ysf_dewhiten(tb, nbits - 16);
    for (int i = 0; i < (nbits - 16) / 8; i++) {
        int b = 0;
        for (int k = 0; k < 8; k++) {
            b = (b << 1) | tb[8 * i + k];
        }
        out_bytes[i] = (uint8_t)b;
    }
    return st;
}

/* One frame's payload (the 360 dibits behind the FICH) by the type the frame is read as: kind = 1 V/D1, 2 V/D2, 4 full-rate voice,
 * 8 full-rate data (ysf_dispatch_payload: several can apply - FI = 0 / 2 with DT = 1 is one full-rate data pass - so a bit mask);
 * dch[2][20] / dch_status[2] / dch_cost[2]: V/D2 -> block 0 = DCH2 (10 bytes); V/D1 -> block 0 = DCH (20 bytes); full-rate data ->
 * both blocks; ambe_d[5][49] + errs2[5] for V/D2.  The voice frames of V/D1 and full-rate voice are not restated. */
int
orc_ysf_payload(const uint8_t p[360], int fi, int dt, uint8_t dch[2][20], uint8_t dch_status[2], uint32_t dch_cost[2],
                uint8_t ambe_d[5][49], uint8_t errs2[5]) {
    int kind = 0;
    uint8_t d[180];
    memset(dch, 0, 40);
    dch_status[0] = dch_status[1] = 0;
    dch_cost[0] = dch_cost[1] = 0;
    memset(ambe_d, 0, 5 * 49);
    memset(errs2, 0, 5);
    if (fi == 1 && dt == 0) {
        kind |= 1;
        for (int i = 0; i < 5; i++) {
            memcpy(d + 36 * i, p + 72 * i, 36);
        }
        dch_status[0] = (uint8_t)orc_ysf_dch(d, 180, dch[0], &dch_cost[0]);
    }
    if (fi == 1 && dt == 2) {
        kind |= 2;
        for (int i = 0; i < 5; i++) {
            memcpy(d + 20 * i, p + 72 * i, 20);
            errs2[i] = (uint8_t)orc_ysf_vd2_voice(p + 72 * i + 20, ambe_d[i]);
        }
        dch_status[0] = (uint8_t)orc_ysf_dch(d, 100, dch[0], &dch_cost[0]);
    }
    if (fi == 1 && dt == 3) {
        kind |= 4;
    }
    if (dt == 1 || fi == 0 || fi == 2) { /* (after a V/D pass of the same frame the reference reads on: not restated - see kind) */
        kind |= 8;
        if (kind == 8) {
            for (int b = 0; b < 2; b++) {
                for (int i = 0; i < 5; i++) {
                    memcpy(d + 36 * i, p + 36 * (2 * i + b), 36);
                }
                dch_status[b] = (uint8_t)orc_ysf_dch(d, 180, dch[b], &dch_cost[b]);
            }
        }
    }
    return kind;
}

/* ---- the voice frames of V/D mode 1 and of full-rate frames, as processMbeFrame() gets them (round 5, last part) -------------------
 *   ysf_ehr()          ysf.c:425-476: 36 voice dibits -> ambe_fr[4][24] through dsd_ambe_2450_dibit_map (the generated schedule,
 *                      ddn_tables_ambe.h); ysf_handle_vd_type1 decodes the first FOUR of a frame's five voice blocks (:683)
 *   full rate          ysf_read_full_rate_imbe_raw / dsd_ysf_unpack_full_rate_imbe (ysf.c:794-802, ysf_frame.c:138-163): 72 dibits ->
 *                      144 bits -> the 24 x 6 interleave (entry r * 24 + c = 12 (c / 2) + (c odd ? (r ^ 1) + 6 : r)) -> imbe_fr[8][23]
 *                      rows 0-3 23 bits, 4-6 15 bits, 7 seven bits, highest column first; checked against the compiled ysf_frame.c
 *   CSD3               ysf_handle_full_rate_voice :824-842: FT = 1 & FN = 0 -> five banks of 36 data dibits (a sixth skipped), two
 *                      voice slots, then ysf_conv_dch over the 180 dibits
 * -> frames [5][184] (AMBE: the first 96 entries, row * 24 + column; IMBE: row * 23 + column), returns how many */
void
orc_ysf_fr_unpack(const uint8_t dibits72[72], uint8_t imbe_fr[8 * 23]) {
    uint8_t raw[144], vch[144];
    for (int j = 0; j < 72; j++) {
        raw[2 * j] = (uint8_t)((dibits72[j] >> 1) & 1);
        raw[2 * j + 1] = (uint8_t)(dibits72[j] & 1);
    }
    for (int j = 0; j < 144; j++) {
        const int r = j / 24, c = j % 24;
        vch[j] = raw[12 * (c >> 1) + ((c & 1) ? ((r ^ 1) + 6) : r)];
    }
    memset(imbe_fr, 0, 8 * 23);
    int k = 0;
    for (int n = 0; n < 4; n++) {
        for (int m = 22; m >= 0; m--) {
            imbe_fr[n * 23 + m] = vch[k++];
        }
    }
    for (int n = 4; n < 7; n++) {
        for (int m = 14; m >= 0; m--) {
            imbe_fr[n * 23 + m] = vch[k++];
        }
    }
    for (int m = 6; m >= 0; m--) {
        imbe_fr[7 * 23 + m] = vch[k++];
    }
}

int
orc_ysf_voice_frames(const uint8_t p[360], int kind, int csd3, uint8_t fr[5][184], uint8_t dch20[20], uint8_t* dch_status, uint32_t* dch_cost) {
    static const uint8_t map[36][4] = DDN_AMBE2450_MAP_INIT;
    memset(fr, 0, 5 * 184);
    if (kind == 1) {
        for (int sf = 0; sf < 4; sf++) {
            for (int i = 0; i < 36; i++) {
                const int d = p[72 * sf + 36 + i] & 3;
                fr[sf][map[i][0] * 24 + map[i][1]] = (uint8_t)(d >> 1);
                fr[sf][map[i][2] * 24 + map[i][3]] = (uint8_t)(d & 1);
            }
        }
        return 4;
    }
    if (kind == 4) {
        const int n = csd3 ? 2 : 5, off = csd3 ? 216 : 0;
        for (int i = 0; i < n; i++) {
            orc_ysf_fr_unpack(p + off + 72 * i, fr[i]);
        }
        if (csd3) {
            *dch_status = (uint8_t)orc_ysf_dch(p, 180, dch20, dch_cost);
        }
        return n;
    }
    return 0;
}
