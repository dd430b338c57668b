/* ddn_host_p25_layout.c — dibit offsets of the P25 Phase 1 frame fields, relative to the first dibit of the frame sync.
 *
 * TIA-102.BAAA framing as the reference's handlers walk it with getDibitSoft(): 24 sync dibits, then the NID (32
 * dibits + the status symbol at frame index 35, src/protocol/p25/phase1/dispatch_p25p1.c:123-143), then the body, with
 * one status symbol after every 35 dibits, i.e. at every frame index = 35 (mod 36) (p25p1_ldu.c:27-39).
 *   TSDU / PDU body: 98-dibit trellis blocks back to back (p25p1_tsbk.c)
 *   LDU1: IMBE 1,2 | LC words 11..8 | IMBE 3 | LC 7..4 | IMBE 4 | LC 3..0 | IMBE 5 | RS parity words 11..8 | IMBE 6 |
 *         parity 7..4 | IMBE 7 | parity 3..0 | IMBE 8 | LSD (16 dibits) | IMBE 9     (p25p1_ldu1.c; each word = 5
 *         dibits = one Hamming(10,6,3) codeword, high bit of each dibit first)
 *   LDU2: same slots, carrying ES words 15..0 (16 data) and parity words 7..0 (p25p1_ldu2.c:211-236)
 *   HDU:  36 Golay(24,6) words of 3 data + 6 parity dibits, sent hex_data[19..0] then hex_parity[15..0]
 *         (p25p1_hdu.c:191-200,252-268; status counter starts at 21 = frame index 57)
 * Pure host code, no device state: the tables are uploaded once and drive k_gather_fields. */
#include "ddn_internal.h"

typedef struct {
    int idx; /* frame index of the next unread dibit */
} walker;

static void
take(walker* w, int n, int32_t* out) {
    int got = 0;
    while (got < n) {
        if (w->idx % 36 == 35) {
            w->idx++; /* status symbol */
            continue;
        }
        if (out) {
            out[got] = w->idx;
        }
        got++;
        w->idx++;
    }
}

int
ddn_p25p1_layout_nid(int32_t out32[32]) {
    walker w = {24};
    take(&w, 32, out32);
    return w.idx; /* 57: first body dibit */
}

int
ddn_p25p1_layout_trellis_block(int block, int32_t out98[98]) {
    if (block < 0 || block > 2 || !out98) {
        return -1;
    }
    walker w = {57};
    for (int b = 0; b < block; b++) {
        take(&w, 98, 0);
    }
    take(&w, 98, out98);
    return w.idx;
}

/* walks an LDU body; word_of_slot maps (slot 0..5, k 0..3) to the output word index */
static int
ldu_walk(int ldu, int32_t* words /* [24][5] */, int32_t* imbe_first /* [9] */, int32_t* imbe_status /* [9] */,
         int32_t* lsd /* [16] */) {
    walker w = {57};
    int v = 0;
    for (int slot = -1; slot < 8; slot++) {
        /* voice frame(s) before this slot's words: IMBE 1 then, per slot, one more */
        if (imbe_first) {
            /* process_IMBE starts at the next unread dibit; if that is a status position the reference's counter
             * shows 35 and the first read steps over it */
            imbe_first[v] = w.idx;
            imbe_status[v] = w.idx % 36;
        }
        v++;
        take(&w, 72, 0);
        if (slot < 0) {
            continue;
        }
        if (slot < 6) {
            for (int k = 0; k < 4; k++) {
                int word;
                if (ldu == 1) { /* hex_data[11..0] over slots 0..2, hex_parity[11..0] over slots 3..5 */
                    word = (slot < 3) ? (11 - 4 * slot - k) : 12 + (11 - 4 * (slot - 3) - k);
                } else { /* hex_data[15..0] over slots 0..3, hex_parity[7..0] over slots 4..5 */
                    word = (slot < 4) ? (15 - 4 * slot - k) : 16 + (7 - 4 * (slot - 4) - k);
                }
                take(&w, 5, words ? words + 5 * word : 0);
            }
        } else if (slot == 6) {
            take(&w, 16, lsd);
        }
        if (slot == 7) {
            break;
        }
    }
    return w.idx;
}

int
ddn_p25p1_layout_ldu_words(int ldu, int32_t out120[120]) {
    if ((ldu != 1 && ldu != 2) || !out120) {
        return -1;
    }
    return ldu_walk(ldu, out120, 0, 0, 0);
}

int
ddn_p25p1_layout_ldu_imbe(int32_t first9[9], int32_t status9[9]) {
    if (!first9 || !status9) {
        return -1;
    }
    return ldu_walk(1, 0, first9, status9, 0);
}

/* HDU: hex3 = [36][3] dibits carrying each word's 6 data bits, par6 = [36][6] dibits carrying its 12 Golay parity bits;
 * word order hex_data[0..19] then hex_parity[0..15] (= the order the Reed-Solomon decoder takes them). */
int
ddn_p25p1_layout_hdu(int32_t hex3[36 * 3], int32_t par6[36 * 6]) {
    if (!hex3 || !par6) {
        return -1;
    }
    walker w = {57};
    for (int seq = 0; seq < 36; seq++) {
        const int word = seq < 20 ? 19 - seq : 20 + (15 - (seq - 20));
        take(&w, 3, hex3 + 3 * word);
        take(&w, 6, par6 + 6 * word);
    }
    return w.idx;
}

/* TDULC (p25p1_tdulc.c:199-207,297): twelve Golay(24,12) words of 6 data + 6 parity dibits, sent dodeca_data[5..0] then
 * dodeca_parity[5..0]; data6 / par6 = [12][6] dibits in the order dodeca_data[0..5], dodeca_parity[0..5]. */
int
ddn_p25p1_layout_tdulc(int32_t data6[72], int32_t par6[72]) {
    if (!data6 || !par6) {
        return -1;
    }
    walker w = {57};
    for (int seq = 0; seq < 12; seq++) {
        const int word = seq < 6 ? 5 - seq : 6 + (5 - (seq - 6));
        take(&w, 6, data6 + 6 * word);
        take(&w, 6, par6 + 6 * word);
    }
    return w.idx;
}

/* the 16 LSD dibits of an LDU (p25p1_ldu1.c:145-183): two (16,8) codewords, each 4 data dibits then 4 parity dibits */
int
ddn_p25p1_layout_ldu_lsd(int32_t out16[16]) {
    if (!out16) {
        return -1;
    }
    return ldu_walk(1, 0, 0, 0, out16);
}
