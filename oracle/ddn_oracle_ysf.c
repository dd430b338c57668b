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
