/*
 * oracle/ddn_oracle_m17.c - TEST INFRASTRUCTURE ONLY: CPU restatement of what the reference's M17 handlers do with the 184 payload
 * symbols behind a frame sync (the consumers of SURVEY 8a rows a17 - the libM17-style K = 5 decoder - and a18 - the NXDN one).
 *
 *   processM17LSF()          src/protocol/m17/m17.c:1395-1408: 184 soft symbols (getDibitAndSoftSymbol) -> per-bit Viterbi costs
 *                            (soft_symbol_to_viterbi_cost, src/core/frames/dsd_dibit.c:1189-1242; llr_to_viterbi_cost :1150-1167)
 *                            against the thresholds as they stand AFTER the frame was read -> de-randomised (cost complemented
 *                            where the randomiser bit is 1) -> de-interleaved (x = 45 i + 92 i^2 mod 368) :1187-1205 ->
 *                            de-punctured with pattern P1, 0x7FFF where a bit was cut :1207-1217 -> viterbi_decode(488 costs)
 *                            (dsd_misc.c:118-139; orc_m17_viterbi_decode) -> bytes 1..30 = the LSF -> CRC16 :1370-1393,1343-1368
 *   processM17STR()          :1122-1176: hard dibits -> bits -> de-randomise -> de-interleave -> 96 LICH bits = four Golay(24,12)
 *                            words (m17_lich_decode_bits, m17_algorithms.c:598-612: Golay_24_12_decode, fec.c) -> 40 LSF bits +
 *                            3-bit chunk counter + 5 reserved (m17_lich_parse_content :562-583); when the LICH decodes:
 *                            M17prepareStream() :1039-1120 = 272 bits de-punctured with P2 (11 of 12 kept, the cut bit 0) ->
 *                            symbol values bit << 1 -> CNXDNConvolution over 148 steps, chain back 144 bits
 *                            (nxdn_convolution.c; orc_nxdn_conv_decode) -> frame number (16) + payload (128)
 *   LSF fields               m17_parse_lsf() src/protocol/m17/m17_parse.c:369-420 (DST 48, SRC 48, TYPE 16, META 112, CRC 16),
 *                            base-40 call signs m17_address_decode_csd() :296-317
 *   CRC                      m17_crc16() m17_algorithms.c:19-35 (poly 0x5935, init 0xFFFF)
 *
 * PARITY STATUS: m17_algorithms.c, m17_tables.c and m17_parse.c compile from the reference's sources into oracle/_ref; the bit
 * shuffles, the CRC, the LICH decode and the call-sign decode here are pinned to them, and the test frames are built with the
 * reference's own encoder (m17_lsf_encode_type1_bits ... m17_frame_build_dibits) - tests/test_oracle_m17.py.  m17.c itself needs
 * <sndfile.h> / codec2: the two handler bodies above are restated.  The randomiser sequence is the M17 specification's 46 bytes.
 */
#include <math.h>
#include <string.h>

#include "ddn_oracle.h"

/* M17 specification, "Randomizer": 46 bytes, most significant bit first (== m17_scramble[], m17_tables.c:16-27) */
static const uint8_t k_m17_rand[46] = {0xD6, 0xB5, 0xE2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xBA, 0x4E, 0x96, 0x90, 0xD8, 0x98, 0xDD, 0x5D,
                                       0x0C, 0xC8, 0x52, 0x43, 0x91, 0x1D, 0xF8, 0x6E, 0x68, 0x2F, 0x35, 0xDA, 0x14, 0xEA, 0xCD, 0x76,
                                       0x19, 0x8D, 0xD5, 0x80, 0xD1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2D, 0x29, 0x78, 0xC3};
int
orc_m17_rand_bit(int i) {
    return (k_m17_rand[i >> 3] >> (7 - (i & 7))) & 1;
}
int
orc_m17_interleave_index(int i) { /* x = (45 i + 92 i^2) mod 368 */
    return (45 * i + 92 * i * i) % 368;
}

uint16_t
orc_m17_crc16(const uint8_t* in, int len) {
    uint32_t crc = 0xFFFFu;
    for (int i = 0; i < len; i++) {
        crc ^= (uint32_t)in[i] << 8;
        for (int j = 0; j < 8; j++) {
            crc <<= 1;
            if (crc & 0x10000u) {
                crc = (crc ^ 0x5935u) & 0xFFFFu;
            }
        }
    }
    return (uint16_t)(crc & 0xFFFFu);
}

static float
min_sq2(float x, float a, float b) {
    const float da = x - a, db = x - b;
    const float d2a = da * da, d2b = db * db;
    return d2a < d2b ? d2a : d2b;
}

/* soft_symbol_to_viterbi_cost(); thr5 = {center, umid, lmid, max, min} */
uint16_t
orc_m17_soft_cost(float symbol, const float thr5[5], int bit) {
    float center = thr5[0], umid = thr5[1], lmid = thr5[2], max_val = thr5[3], min_val = thr5[4];
    if (!(min_val < lmid && lmid < center && center < umid && umid < max_val)) {
        float span = max_val - min_val;
        if (span < 1e-3f) {
            span = 2.0f;
        }
        const float half = span * 0.5f;
        min_val = center - half;
        max_val = center + half;
        lmid = center - (span / 6.0f);
        umid = center + (span / 6.0f);
    }
    const float n3 = 0.5f * (min_val + lmid), n1 = 0.5f * (lmid + center), p1 = 0.5f * (center + umid), p3 = 0.5f * (umid + max_val);
    float sigma = (max_val - min_val) / 6.0f;
    if (sigma < 1e-3f) {
        sigma = 1e-3f;
    }
    const float inv_2sigma2 = 0.5f / (sigma * sigma);
    float d0, d1;
    if ((bit & 1) == 0) {
        d0 = min_sq2(symbol, p1, p3);
        d1 = min_sq2(symbol, n1, n3);
    } else {
        d0 = min_sq2(symbol, n1, p1);
        d1 = min_sq2(symbol, n3, p3);
    }
    const float llr = (d1 - d0) * inv_2sigma2;
    if (llr >= 16.0f) {
        return 0;
    }
    if (llr <= -16.0f) {
        return 65535;
    }
    const float pr1 = 1.0f / (1.0f + expf(llr));
    long q = lrintf(pr1 * 65535.0f);
    q = q < 0 ? 0 : (q > 65535 ? 65535 : q);
    return (uint16_t)q;
}

/* m17_soft_bits_from_symbols() + m17_soft_depuncture_p1(): 184 symbols -> 488 costs */
void
orc_m17_lsf_costs(const float* sym184, const float thr5[5], uint16_t cost488[488]) {
    uint16_t rnd[368], il[368], bits[368];
    for (int i = 0; i < 184; i++) {
        rnd[2 * i] = orc_m17_soft_cost(sym184[i], thr5, 0);
        rnd[2 * i + 1] = orc_m17_soft_cost(sym184[i], thr5, 1);
    }
    for (int i = 0; i < 368; i++) {
        il[i] = orc_m17_rand_bit(i) ? (uint16_t)(0xFFFFu - rnd[i]) : rnd[i];
    }
    for (int i = 0; i < 368; i++) {
        bits[i] = il[orc_m17_interleave_index(i)];
    }
    int k = 0;
    for (int i = 0; i < 488; i++) { /* P1: 61 entries, every fourth of {1,1,0,1} cut except that the pattern ends 1,1 */
        const int j = i % 61;
        const int keep = (j == 60) ? 1 : ((j & 3) != 2);
        cost488[i] = keep ? bits[k++] : 0x7FFFu;
    }
}

/* m17_decode_lsf_soft_bits(): -> the 30 LSF bytes, returns 1 when the CRC16 over the first 28 equals the last two */
int
orc_m17_lsf_decode(const uint16_t cost488[488], uint8_t lsf30[30], uint32_t* path_cost) {
    uint8_t by[32];
    memset(by, 0, sizeof(by));
    const uint32_t c = orc_m17_viterbi_decode(by, cost488, 488);
    if (path_cost) {
        *path_cost = c;
    }
    memcpy(lsf30, by + 1, 30);
    const uint16_t ext = (uint16_t)((lsf30[28] << 8) | lsf30[29]);
    return orc_m17_crc16(lsf30, 28) == ext;
}

/* m17_payload_decode_bits(): 368 received bits -> de-randomised, de-interleaved */
void
orc_m17_payload_bits(const uint8_t* dibits184, uint8_t bits368[368]) {
    uint8_t rnd[368];
    for (int i = 0; i < 184; i++) {
        rnd[2 * i] = (dibits184[i] >> 1) & 1;
        rnd[2 * i + 1] = dibits184[i] & 1;
    }
    for (int i = 0; i < 368; i++) {
        const int x = orc_m17_interleave_index(i);
        bits368[i] = (uint8_t)((rnd[x] ^ orc_m17_rand_bit(x)) & 1);
    }
}

/* processM17STR() up to the payload: returns 0 when all four LICH words decode and the chunk counter is < 6 (else -1, and the
 * payload is not decoded - as the reference).  lich6 = the 48 decoded content bits packed; fn_payload18 = frame number (2 bytes,
 * big endian) + the 16 payload bytes. */
int
orc_m17_str_decode(const uint8_t* dibits184, uint8_t lich6[6], int* lich_cnt, uint8_t fn_payload18[18]) {
    uint8_t bits[368], content[48];
    orc_m17_payload_bits(dibits184, bits);
    int err = 0;
    for (int b = 0; b < 4; b++) {
        uint8_t rx[24];
        memcpy(rx, bits + 24 * b, 24);
        if (!orc_golay_dmr_decode(24, rx)) {
            err = -1;
        }
        memcpy(content + 12 * b, rx, 12);
    }
    const int cnt = (content[40] << 2) | (content[41] << 1) | content[42];
    if (cnt >= 6) {
        err = -1;
    }
    memset(lich6, 0, 6);
    for (int i = 0; i < 48; i++) {
        lich6[i >> 3] |= (uint8_t)(content[i] << (7 - (i & 7)));
    }
    *lich_cnt = cnt;
    memset(fn_payload18, 0, 18);
    if (err != 0) {
        return err;
    }
    /* M17prepareStream(): 272 bits -> 25 groups of 11 kept + 1 cut (P2), the cut bit reads 0 */
    uint8_t punc[275], depunc[300], sym[296];
    memset(punc, 0, sizeof(punc));
    memcpy(punc, bits + 96, 272);
    int k = 0, x = 0;
    for (int g = 0; g < 25; g++) {
        for (int j = 0; j < 11; j++) {
            depunc[k++] = punc[x++];
        }
        depunc[k++] = 0;
    }
    for (int i = 0; i < 296; i++) {
        sym[i] = (uint8_t)(depunc[i] << 1);
    }
    uint16_t metrics[16];
    memset(metrics, 0, sizeof(metrics));
    orc_nxdn_conv_decode(sym, NULL, 148, metrics, fn_payload18, 144);
    return 0;
}

/* m17_address_decode_csd(): base-40 call sign of a standard address, least significant character first; returns 0, or -2 when the
 * address is not a standard one (0, or above 40^9 - 1) */
int
orc_m17_callsign(uint64_t address, char out10[10]) {
    static const char alphabet[] = " ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-/.";
    memset(out10, 0, 10);
    if (address == 0 || address > 262143999999999ull) {
        return -2;
    }
    for (int i = 0; i < 9 && address != 0; i++) {
        out10[i] = alphabet[address % 40];
        address /= 40;
    }
    return 0;
}
