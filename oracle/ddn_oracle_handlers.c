/*
 * oracle/ddn_oracle_handlers.c - TEST INFRASTRUCTURE ONLY: what the reference's protocol handlers do to the receive loop,
 * restated as per-symbol state machines.  A handler is entered when getFrameSync() returns a sync type; every symbol it
 * reads is read "in frame" (getSymbol(opts, state, 1): no timing slip, the sync-time clip, the in-frame slicer), and the
 * moment it returns the loop hunts again.  How many symbols a handler reads depends on what it decodes, so this is part of
 * the dibit stream's definition, not protocol decoration:
 *
 *   P25 Phase 1   processFrame -> dsd_dispatch_handle_p25p1: NID 32 dibits + 1 status symbol
 *                 (src/engine/dispatch/dispatch_p25p1.c:86-143), p25p1_nid_decode -> DUID (:206-225), then per DUID
 *                 (:391-403): HDU 339 symbols (p25p1_hdu.c:54-70,238-266,408), LDU1 / LDU2 807 (p25p1_ldu1.c:120-220,
 *                 p25p1_ldu2.c), TDU 15 (p25p1_tdu.c:35-52), TDULC 159 (p25p1_tdulc.c:181-253), TSDU 101 per block until
 *                 the last-block flag of the decoded block, at most 3 (p25p1_tsbk.c:117-161,1051-1072), PDU header block
 *                 then blks + 1 blocks (p25p1_mdpu.c:177-198,270-307), anything else 0 (:371-386).
 *   DMR (-fs)     dsd_dispatch_handle_dmr (src/engine/dispatch/dispatch_dmr.c:126-158): BS data sync -> dmr_data_sync()
 *                 (src/protocol/dmr/dmr_data.c:117-343: TACT Hamming(7,4) on the cached CACH, slot type Golay(20,8) after
 *                 5 live dibits, 49 more, then skipDibit(66)); BS voice sync -> dmrBSBootstrap() + dmrBS()
 *                 (src/protocol/dmr/dmr_bs.c:697-948: 54 live dibits, then 144 per burst with the TACT / repeated-carrier /
 *                 sync-word / EMB QR(16,7,6) / colour-code-gate decisions of :196-470,555-620,796-880), the gate itself
 *                 src/protocol/dmr/dmr_confidence.c.
 *   NXDN          nxdn_frame() (src/protocol/nxdn/nxdn_frame.c:181-233,592-640): 8 LICH dibits, parity / profile check,
 *                 then 174 more; a rejected LICH clears lastsynctype.
 *
 * The decoders called here (BCH / NID ladder, half-rate list decoder, CRC16, Hamming(7,4), Golay(20,8), QR(16,7,6)) are
 * the restatements already pinned to the compiled reference (ddn_oracle_block.c, ddn_oracle_fec.c, ddn_oracle_fec3.c).
 * The control flow has no compiled counterpart (the handlers pull the whole engine in): PARITY of the flow is anchored
 * on the reference's full-chain known answers (tests/test_oracle_rx4.py, tests/test_real_capture.py).
 */
#include <string.h>

#include "ddn_oracle.h"

static void
ev_push_data(orc_hevents* e, long pos, int kind, int a, int b, int c, const int32_t* data4) {
    if (!e) {
        return;
    }
    if (e->n < ORC_HEV_MAX) {
        orc_hevent* v = &e->ev[e->n];
        v->pos = (int32_t)pos;
        v->kind = (int16_t)kind;
        v->a = (int16_t)a;
        v->b = (int16_t)b;
        v->c = (int16_t)c;
        for (int k = 0; k < 4; k++) {
            v->data[k] = data4 ? data4[k] : 0;
        }
    }
    e->n++;
}

static void
ev_push(orc_hevents* e, long pos, int kind, int a, int b, int c) {
    ev_push_data(e, pos, kind, a, b, c, 0);
}

static void
block_words(const uint8_t by[12], int crc_ok, int sel, int block, int32_t out4[4]) {
    for (int w = 0; w < 3; w++) {
        out4[w] = (int32_t)((uint32_t)by[4 * w] | ((uint32_t)by[4 * w + 1] << 8) | ((uint32_t)by[4 * w + 2] << 16)
                            | ((uint32_t)by[4 * w + 3] << 24));
    }
    out4[3] = (crc_ok & 1) | ((sel & 0xFF) << 8) | (block << 16);
}

/* ------------------------------------------------------------------------------------------------ P25 Phase 1 ---- */
enum { P25H_IDLE = 0, P25H_NID, P25H_BODY, P25H_TSBK, P25H_MPDU };

void
orc_p25h_init(orc_p25h* h, int erasure_threshold) {
    memset(h, 0, sizeof(*h));
    h->threshold = erasure_threshold;
}

void
orc_p25h_no_carrier(orc_p25h* h) { /* engine.c:1889 state->nac = 0; p2_cc stays */
    h->nac = 0;
}

/* entered on a P25p1 sync; returns 1 (the NID is always read) */
int
orc_p25h_begin(orc_p25h* h) {
    h->phase = P25H_NID;
    h->idx = 0;
    h->duid = 0xFF;
    return 1;
}

static int
llr_rel(int llr) { /* p25p1_llr_reliability(), dispatch_p25p1.c:59-66 */
    int v = llr < 0 ? -llr : llr;
    return v > 255 ? 255 : v;
}

static int
nac_valid_observed(long nac) { /* p25p1_valid_observed_nac */
    return nac > 0 && nac <= 0xFFF && nac != 0xFFF;
}

static int
body_symbols(int duid) {
    switch (duid) {
        case 0x0: return 339; /* 36 hex words x 9 dibits, 9 status symbols on the way (count from 21), 5 + 1 trailing */
        case 0x5:
        case 0xA: return 807; /* 864 - 24 - 33 */
        case 0x3: return 15;  /* 14 null dibits + status */
        case 0xF: return 159; /* 144 + 10 null dibits, 4 status symbols, trailing status */
        default: return 0;
    }
}

/* block reader shared by processTSBK and p25_mpdu_read_repetition: returns 1 when the symbol is a data dibit */
static int
block_take(orc_p25h* h, int d, int l0, int l1) {
    int data = 0;
    if ((h->skipdibit / 36) == 0) {
        if (h->k < 98) {
            h->dib[h->k] = (uint8_t)d;
            h->llr[2 * h->k] = (int16_t)l0;
            h->llr[2 * h->k + 1] = (int16_t)l1;
        }
        h->k++;
        data = 1;
    } else {
        h->skipdibit = 0;
    }
    h->skipdibit++;
    return data;
}

/* tsbk_decode_repetition_bytes / p25_mpdu_decode_r12_block(block 0): list-8 decode, first CRC16-clean candidate else the best */
static int
half_rate_select(const orc_p25h* h, uint8_t out12[12], int* crc_ok) {
    uint8_t cand[8][12];
    uint32_t metric[8];
    const int n = orc_p25_12_soft_llr_list(h->llr, &cand[0][0], metric, 8);
    int sel = 0;
    *crc_ok = 0;
    if (n > 0) {
        for (int c = 0; c < n; c++) {
            if (orc_p25_crc16_ok(cand[c], 10) == 0) {
                sel = c;
                *crc_ok = 1;
                break;
            }
        }
        memcpy(out12, cand[sel], 12);
        return sel;
    }
    (void)orc_p25_12_soft_llr(h->llr, out12);
    *crc_ok = (orc_p25_crc16_ok(out12, 10) == 0);
    return -1;
}

/* one in-frame symbol: dibit + its two LLRs.  Returns 1 = the handler reads on, 0 = it has returned. */
int
orc_p25h_symbol(orc_p25h* h, long pos, int d, int l0, int l1, orc_hevents* ev) {
    switch (h->phase) {
        case P25H_NID: {
            const int i = h->idx++;
            if (i != 11) { /* dibit 11 is the status symbol inside the NID */
                const int b = (i < 11) ? 2 * i : 2 * (i - 1);
                if (i < 32) {
                    h->bch[b] = (uint8_t)((d >> 1) & 1);
                    h->bch_rel[b] = (uint8_t)llr_rel(l0);
                    h->bch[b + 1] = (uint8_t)(d & 1);
                    h->bch_rel[b + 1] = (uint8_t)llr_rel(l1);
                } else {
                    h->bch[62] = (uint8_t)((d >> 1) & 1);
                    h->bch_rel[62] = (uint8_t)llr_rel(l0);
                    h->parity = d & 1;
                    h->parity_rel = llr_rel(l1);
                }
            }
            if (i < 32) {
                return 1;
            }
            /* p25p1_decode_nid_and_duid(): observed NAC = state->nac when valid, else p2_cc */
            const int observed = nac_valid_observed(h->nac) ? h->nac : (nac_valid_observed(h->p2_cc) ? h->p2_cc : 0);
            int out4[4];
            orc_p25p1_nid_decode(h->bch, h->bch_rel, observed, h->parity, h->parity_rel, h->threshold, out4);
            int duid = 0xFF;
            if (out4[0] > 0) {
                const int new_nac = out4[1];
                const int valid = new_nac != 0 && new_nac != 0xFFF;
                if (new_nac != h->nac && valid) {
                    h->nac = new_nac;
                    h->p2_cc = new_nac;
                }
                duid = out4[2];
            }
            h->duid = duid;
            {
                const int32_t d4[4] = {out4[0], out4[1], out4[2], out4[3]};
                ev_push_data(ev, pos, ORC_HEV_P25_NID, out4[0], out4[1], duid, d4);
            }
            if (duid == 0x7 || duid == 0xC) {
                h->phase = (duid == 0x7) ? P25H_TSBK : P25H_MPDU;
                h->block = 0;
                h->end = 3; /* TSBK_MAX_BLOCKS; p25_mpdu_context_init() end = 3 */
                h->skipdibit = 36 - 14;
                h->idx = 0;
                h->k = 0;
                return 1;
            }
            h->left = body_symbols(duid);
            if (h->left <= 0) {
                h->phase = P25H_IDLE;
                return 0;
            }
            h->phase = P25H_BODY;
            return 1;
        }
        case P25H_BODY:
            if (--h->left <= 0) {
                h->phase = P25H_IDLE;
                return 0;
            }
            return 1;
        case P25H_TSBK: {
            (void)block_take(h, d, l0, l1);
            if (++h->idx < 101) {
                return 1;
            }
            uint8_t by[12];
            int crc_ok;
            const int sel = half_rate_select(h, by, &crc_ok);
            const int last = (by[0] >> 7) & 1;
            {
                int32_t d4[4];
                block_words(by, crc_ok, sel, h->block, d4);
                ev_push_data(ev, pos, ORC_HEV_P25_TSBK, h->block, crc_ok | (by[1] << 8), (last << 8) | (sel & 0xFF), d4);
            }
            h->block++;
            h->idx = 0;
            h->k = 0;
            if (last || h->block >= 3) {
                h->phase = P25H_IDLE;
                return 0;
            }
            return 1;
        }
        case P25H_MPDU: {
            (void)block_take(h, d, l0, l1);
            h->idx++;
            if (h->k < 98 && h->idx < 101) {
                return 1;
            }
            if (h->block == 0) {
                uint8_t by[12];
                int crc_ok;
                const int sel = half_rate_select(h, by, &crc_ok);
                /* p25_mpdu_update_header_from_first_block(): opts->aggressive_framesync = 1 (dsd_init.c:246) keeps the
                 * defaults when the header CRC fails */
                if (crc_ok) {
                    const int sap = by[1] & 0x3F, blks = by[6] & 0x7F;
                    h->end = blks + 1;
                    if ((sap == 61 || sap == 63) && blks > 10) {
                        h->end = 4;
                    }
                }
                {
                    int32_t d4[4];
                    block_words(by, crc_ok, sel, 0, d4);
                    ev_push_data(ev, pos, ORC_HEV_P25_MPDU, crc_ok, h->end, by[0], d4);
                }
            }
            h->block++;
            h->idx = 0;
            h->k = 0;
            if (h->block >= h->end) {
                h->phase = P25H_IDLE;
                return 0;
            }
            return 1;
        }
        default: return 0;
    }
}

/* p25_mpdu_finalize_header() (src/protocol/p25/phase1/p25p1_mdpu.c:381-410) with its two helpers, p25_mpdu_try_combined_header()
 * (:336-360; saturating_llr_add :36-45) and p25_mpdu_rebuild_header_from_majority() (:362-379).
 *   rep_bytes [3][12]  the repetitions as p25_mpdu_collect_blocks() stored them (hdr_rep_bytes: block 0 = the CRC16-selected candidate,
 *                      blocks 1 / 2 = the list decoder's first candidate, :233-243,250-261)
 *   rep_llr   [3][196] the blocks' LLRs (hdr_rep_llr)
 *   hdr_reps           1..3 (:382-383: 1 when the header announces data blocks, else min(end, 3))
 * out12 = ctx->mpdu_byte[0..11] afterwards.  Returns how it got there: 0 / 1 / 2 = that repetition's CRC16 held, 32 = the summed LLRs'
 * list decode (first candidate with a good CRC16), 64 = bitwise majority with a good CRC16, 64 | 16 = majority, CRC16 still bad
 * (ctx->err[0] != 0). */
int
orc_p25_mpdu_finalize_header(const uint8_t* rep_bytes, const int16_t* rep_llr, int hdr_reps, uint8_t out12[12]) {
    int selected = -1;
    for (int rep = 0; rep < hdr_reps; rep++) {
        if (orc_p25_crc16_ok(rep_bytes + 12 * rep, 10) == 0) {
            selected = rep;
            break;
        }
    }
    if (selected >= 0) {
        memcpy(out12, rep_bytes + 12 * selected, 12);
        return selected;
    }
    memcpy(out12, rep_bytes, 12); /* (ctx->mpdu_byte[0..11] holds block 0 as stored, :263-277) */
    if (hdr_reps > 1) {
        int16_t comb[196];
        for (int bit = 0; bit < 196; bit++) {
            int acc = 0;
            for (int rep = 0; rep < hdr_reps; rep++) {
                acc += rep_llr[196 * rep + bit];
                acc = acc > 32767 ? 32767 : (acc < -32768 ? -32768 : (int16_t)acc);
            }
            comb[bit] = (int16_t)acc;
        }
        uint8_t cand[8][12];
        uint32_t metric[8];
        const int n = orc_p25_12_soft_llr_list(comb, &cand[0][0], metric, 8);
        for (int c = 0; c < n; c++) {
            if (orc_p25_crc16_ok(cand[c], 10) == 0) {
                memcpy(out12, cand[c], 12);
                return 32;
            }
        }
    }
    const int thresh = (hdr_reps >= 2) ? ((hdr_reps + 1) / 2) : 1;
    uint8_t maj[12];
    memset(maj, 0, sizeof(maj));
    for (int bit = 0; bit < 96; bit++) {
        int sum = 0;
        for (int rep = 0; rep < hdr_reps; rep++) {
            sum += (rep_bytes[12 * rep + (bit >> 3)] >> (7 - (bit & 7))) & 1;
        }
        if (sum >= thresh) {
            maj[bit >> 3] |= (uint8_t)(0x80 >> (bit & 7));
        }
    }
    memcpy(out12, maj, 12);
    return orc_p25_crc16_ok(maj, 10) == 0 ? 64 : (64 | 16);
}

/* ---------------------------------------------------------------------------------------------------- NXDN ---- */
static const uint8_t k_nxdn_lich_ok[] = {0x01, 0x05, 0x28, 0x29, 0x49, 0x2E, 0x2F, 0x4E, 0x4F, 0x32, 0x33, 0x52, 0x53,
                                         0x34, 0x35, 0x54, 0x55, 0x36, 0x37, 0x56, 0x57, 0x20, 0x21, 0x30, 0x31, 0x40,
                                         0x41, 0x50, 0x51, 0x38, 0x39, 0x46, 0x08, 0x48, 0x4A, 0x76, 0x77, 0x75, 0x72,
                                         0x73, 0x70, 0x71, 0x6E, 0x6F, 0x68, 0x69, 0x62, 0x63, 0x60, 0x61};

void
orc_nxdnh_init(orc_nxdnh* h) {
    memset(h, 0, sizeof(*h));
}

int
orc_nxdnh_begin(orc_nxdnh* h) {
    h->idx = 0;
    h->lich = 0;
    return 1;
}

/* returns 1 = reads on, 0 = returned; *bad_sync = 1 when the LICH was rejected (nxdn_mark_bad_sync: lastsynctype NONE) */
int
orc_nxdnh_symbol(orc_nxdnh* h, long pos, int d, int* bad_sync, orc_hevents* ev) {
    *bad_sync = 0;
    const int i = h->idx++;
    if (i < 8) {
        /* nxdn_descramble_with_seed(.., 8, 228): PN9 from seed 228, dibit ^= 2 where the sequence is 1 */
        static const uint8_t pn8[8] = {0, 0, 1, 0, 0, 1, 1, 1};
        const int dd = d ^ (pn8[i] << 1);
        h->lich |= ((dd >> 1) & 1) << (7 - i);
        if (i < 7) {
            return 1;
        }
        const int full = h->lich;
        const int rx_par = full & 1;
        int par = ((full >> 7) + (full >> 6) + (full >> 5) + (full >> 4)) & 1;
        const int lich = full >> 1;
        if (lich == 0x08 || lich == 0x4A || lich == 0x48 || lich == 0x46) {
            par = ((full >> 7) + (full >> 6) + (full >> 5) + (full >> 4) + (full >> 3) + (full >> 2) + (full >> 1)) & 1;
        }
        int ok = (rx_par == par);
        if (ok) { /* direction check passes without trunking; then the profile table */
            ok = 0;
            for (size_t k = 0; k < sizeof(k_nxdn_lich_ok); k++) {
                if (k_nxdn_lich_ok[k] == lich) {
                    ok = 1;
                    break;
                }
            }
        }
        ev_push(ev, pos, ORC_HEV_NXDN_LICH, ok, lich, rx_par == par);
        if (!ok) {
            *bad_sync = 1;
            return 0;
        }
        return 1;
    }
    return i < 181; /* 8 + 174 */
}

/* ----------------------------------------------------------------------------------------------------- DMR ---- */
enum { DMRH_IDLE = 0, DMRH_DATA_SUFFIX, DMRH_DATA_SECOND, DMRH_SKIP66, DMRH_BOOT54, DMRH_BURST, DMRH_FIXED };

static const uint8_t k_cach_il[24] = {0, 7, 8, 9, 1, 10, 11, 12, 2, 13, 14, 15, 3, 16, 4, 17, 18, 19, 5, 20, 21, 22, 6, 23};
static const char k_bs_data[25] = "313333111331131131331131";
static const char k_bs_voice[25] = "131111333113313313113313";

static void
conf_clear_voice(orc_dmrh* h) {
    memset(h->vsync_seen, 0, sizeof(h->vsync_seen));
    memset(h->vopen, 0, sizeof(h->vopen));
    memset(h->vcount, 0, sizeof(h->vcount));
}

static void
conf_reset(orc_dmrh* h) { /* dmr_confidence_reset() */
    h->locked = 0;
    h->conf_cc = 16;
    h->cand_cc = 16;
    h->cand_count = 0;
    h->mismatch = 0;
    conf_clear_voice(h);
}

static void
conf_reset_slot(orc_dmrh* h, int slot) {
    h->vsync_seen[slot] = 0;
    h->vopen[slot] = 0;
    h->vcount[slot] = 0;
}

enum { CONF_REJECT = 0, CONF_PENDING = 1, CONF_LOCKED = 2 };

static int
conf_observe(orc_dmrh* h, int cc, int may_lock) { /* dmr_confidence_observe_cc(), dmr_confidence.c:52-107 */
    if (cc < 0 || cc > 15) {
        return CONF_REJECT;
    }
    if (h->locked) {
        if (h->conf_cc == cc) {
            h->cand_cc = 16;
            h->cand_count = 0;
            return CONF_LOCKED;
        }
        if (h->mismatch < 255) {
            h->mismatch++;
        }
        if (h->cand_cc != cc) {
            h->cand_cc = cc;
            h->cand_count = 1;
        } else if (h->cand_count < 255) {
            h->cand_count++;
        }
        if (h->cand_count >= 4) {
            h->conf_cc = cc;
            h->dmr_color_code = cc;
            h->cand_cc = 16;
            h->cand_count = 0;
            h->mismatch = 0;
            conf_clear_voice(h);
            return CONF_LOCKED;
        }
        return CONF_REJECT;
    }
    if (h->cand_cc != cc) {
        h->cand_cc = cc;
        h->cand_count = 1;
    } else if (h->cand_count < 255) {
        h->cand_count++;
    }
    if (may_lock && h->cand_count >= 2) {
        h->locked = 1;
        h->conf_cc = cc;
        h->dmr_color_code = cc;
        return CONF_LOCKED;
    }
    return CONF_PENDING;
}

static void
conf_note_voice_sync(orc_dmrh* h, int slot) {
    h->vsync_seen[slot] = 1;
    if (!h->vopen[slot]) {
        h->vcount[slot] = 0;
    }
}

static int
conf_note_voice_burst(orc_dmrh* h, int slot, int cc) { /* dmr_confidence.c:120-148 */
    if (!h->vsync_seen[slot] && !h->vopen[slot]) {
        return CONF_PENDING;
    }
    const int was_locked = h->locked != 0;
    const int r = conf_observe(h, cc, 1);
    if (r != CONF_LOCKED) {
        return r;
    }
    if (!was_locked && h->vsync_seen[slot]) {
        h->vcount[slot] = 2;
    } else if (h->vcount[slot] < 255) {
        h->vcount[slot]++;
    }
    if (h->vcount[slot] >= 2) {
        h->vopen[slot] = 1;
        return CONF_LOCKED;
    }
    return CONF_PENDING;
}

static int
conf_note_data_burst(orc_dmrh* h, int cc) {
    if (cc < 0 || cc > 15) {
        return CONF_REJECT;
    }
    return conf_observe(h, cc, h->locked ? 0 : 1);
}

void
orc_dmrh_init(orc_dmrh* h) {
    memset(h, 0, sizeof(*h));
    h->dmr_color_code = 16; /* dsd_init.c:1099 */
    conf_reset(h);
}

void
orc_dmrh_no_carrier(orc_dmrh* h) { /* engine.c:1856 */
    conf_reset(h);
}

static int
tact_decode(const uint8_t* pay, uint8_t tact[7], uint8_t cach[24]) {
    for (int i = 0; i < 12; i++) {
        const int d = pay[i];
        cach[k_cach_il[2 * i]] = (uint8_t)((d >> 1) & 1);
        cach[k_cach_il[2 * i + 1]] = (uint8_t)(d & 1);
    }
    memcpy(tact, cach, 7);
    return orc_hamming_7_4_decode(tact) ? 1 : 0;
}

static int
sync_is(const uint8_t* pay66, const char* word) {
    for (int i = 0; i < 24; i++) {
        if ((char)((pay66[i] | 1) + 48) != word[i]) {
            return 0;
        }
    }
    return 1;
}

/* dmr_data_sync() on h->pay[0..143] once the slot type is complete (all 144 dibits present when stereo = 1; 0..94 when
 * called from the dispatcher).  Returns SlotTypeOk. */
static int
data_slot_type(orc_dmrh* h, long pos, orc_hevents* ev, int* burst_out) {
    uint8_t st[20];
    for (int i = 0; i < 5; i++) {
        st[2 * i] = (uint8_t)((h->pay[61 + i] >> 1) & 1);
        st[2 * i + 1] = (uint8_t)(h->pay[61 + i] & 1);
        st[10 + 2 * i] = (uint8_t)((h->pay[90 + i] >> 1) & 1);
        st[11 + 2 * i] = (uint8_t)(h->pay[90 + i] & 1);
    }
    if (!orc_golay_dmr_decode(20, st)) {
        ev_push(ev, pos, ORC_HEV_DMR_DATA, 0, -1, -1);
        return 0;
    }
    const int cc = (st[0] << 3) | (st[1] << 2) | (st[2] << 1) | st[3];
    const int burst = (st[4] << 3) | (st[5] << 2) | (st[6] << 1) | st[7];
    h->color_code = cc;
    h->conf_reject = 0;
    h->conf_pending = 0;
    const int c = conf_note_data_burst(h, cc); /* dmr_ms_mode == 0 on the BS path */
    if (c == CONF_REJECT) {
        h->conf_reject = 1;
    } else if (c != CONF_LOCKED && burst != 9) {
        h->conf_pending = 1;
    }
    *burst_out = burst;
    ev_push(ev, pos, ORC_HEV_DMR_DATA, 1, cc, burst | (h->conf_reject << 8) | (h->conf_pending << 9));
    return 1;
}

/* dmr_data_dispatch_burst(): the burst handler prints the colour code the gate holds (dmr_dburst.c:200-211) */
static void
data_dispatch(orc_dmrh* h, long pos, orc_hevents* ev) {
    if (h->conf_reject || h->conf_pending) {
        return;
    }
    ev_push(ev, pos, ORC_HEV_DMR_CC_PRINT, h->dmr_color_code, 0, h->currentslot);
}

/* entered on a DMR sync.  cls = ORC_DMR_BS_DATA / ORC_DMR_BS_VOICE; pre90 / rel90 = the payload history the reference
 * reads at state->dmr_payload_p - 90 (oldest first, the last 24 are the sync).  Returns 1 when live symbols follow. */
int
orc_dmrh_begin(orc_dmrh* h, long pos, int cls, const uint8_t* pre90, const uint8_t* rel90, orc_hevents* ev) {
    memcpy(h->pay, pre90, 90);
    memcpy(h->rel, rel90, 90);
    uint8_t tact[7], cach[24];
    if (cls == ORC_DMR_BS_DATA) { /* dmr_handle_other_data(): state->dmr_stereo = 0, dmr_data_sync() */
        h->stereo = 0;
        if (!tact_decode(h->pay, tact, cach)) {
            ev_push(ev, pos, ORC_HEV_DMR_DATA, 0, -2, -1);
            h->phase = DMRH_SKIP66; /* dmr_data_finalize(): skipDibit(12 + 49 + 5) */
            h->left = 66;
            return 1;
        }
        h->currentslot = tact[1];
        h->phase = DMRH_DATA_SUFFIX;
        h->idx = 90;
        return 1;
    }
    /* dmrBSBootstrap(), dmr_bs.c:909-946 */
    h->stereo = 1;
    int tact_ok = tact_decode(h->pay, tact, cach), sync_ok = 1;
    if (tact_ok) {
        h->currentslot = tact[1];
        conf_note_voice_sync(h, tact[1]);
        sync_ok = sync_is(h->pay + 66, k_bs_voice);
    }
    if (!tact_ok || !sync_ok) {
        conf_reset(h);
        ev_push(ev, pos, ORC_HEV_DMR_VOICE_END, 1, tact_ok, sync_ok);
        h->phase = DMRH_IDLE;
        return 0;
    }
    h->boot_slot = tact[1];
    h->phase = DMRH_BOOT54;
    h->idx = 90;
    return 1;
}

static void
bs_finalize(orc_dmrh* h, long pos, orc_hevents* ev) { /* finalize_dmr_bs() */
    ev_push(ev, pos, ORC_HEV_DMR_VOICE_END, 0, h->tact_okay, h->emb_ok);
    h->emb_err[0] = h->emb_err[1] = 0;
    conf_reset(h);
    h->phase = DMRH_IDLE;
}

/* the decisions of process_dmr_bs_iteration() after all 144 dibits of a burst are in; returns 1 = next burst, 0 = END */
static int
bs_burst_done(orc_dmrh* h, long pos, orc_hevents* ev) {
    const int slot = h->internalslot;
    uint8_t sd[48];
    for (int i = 0; i < 24; i++) {
        sd[2 * i] = (uint8_t)((h->pay[66 + i] >> 1) & 1);
        sd[2 * i + 1] = (uint8_t)(h->pay[66 + i] & 1);
    }
    uint8_t emb[16];
    for (int i = 0; i < 8; i++) {
        emb[i] = sd[i];
        emb[i + 8] = sd[i + 40];
    }
    const int is_voice = sync_is(h->pay + 66, k_bs_voice), is_data = sync_is(h->pay + 66, k_bs_data);
    /* the VC the sync segment was read under (read_dmr_bs_sync_segment(), dmr_bs.c:161-180: 2..6 files the 48 bits as that burst's
     * embedded signalling) */
    const int vc_read = slot == 0 ? h->vc1 : h->vc2;
    if (is_voice) { /* note_dmr_bs_voice_sync() */
        if (slot == 0) {
            h->vc1 = 1;
            h->emb_err[0] = 0;
        } else {
            h->vc2 = 1;
            h->emb_err[1] = 0;
        }
        conf_note_voice_sync(h, slot);
    }
    int action; /* 0 continue (falls to SKIP), 1 SKIP, 2 END */
    if (is_data) { /* handle_dmr_bs_data_sync(): dmr_data_sync() on the stereo payload */
        if (slot == 0) {
            h->vc1 = 7;
        } else {
            h->vc2 = 7;
        }
        uint8_t tact[7], cach[24];
        if (tact_decode(h->pay, tact, cach)) {
            h->currentslot = tact[1];
            int burst = 0;
            if (data_slot_type(h, pos, ev, &burst)) {
                data_dispatch(h, pos, ev);
            }
        }
        h->skipcount++;
        action = 1;
    } else {
        action = 0;
        /* handle_dmr_bs_frame_sync_miss() */
        if (slot == 0 && h->vc1 > 6) {
            h->vc1++;
            action = (h->vc1 > 13) ? 2 : 1;
        } else if (slot == 1 && h->vc2 > 6) {
            h->vc2++;
            action = (h->vc2 > 13) ? 2 : 1;
        }
        if (action == 0) { /* process_dmr_bs_voice_burst() */
            int cc = 25;
            h->emb_ok = orc_qr_16_7_6_decode(emb) ? 1 : 0;
            int ended = 0;
            if (h->emb_ok) {
                h->emb_err[slot] = 0;
                cc = (emb[0] << 3) | (emb[1] << 2) | (emb[2] << 1) | emb[3];
                h->color_code = cc;
            } else if (!is_voice) {
                if (h->emb_err[slot] < 0xFF) {
                    h->emb_err[slot]++;
                }
                if (h->emb_err[slot] >= 2) {
                    ended = 1;
                }
            } else {
                h->emb_err[slot] = 0;
            }
            if (!ended) {
                int open = h->vopen[slot] != 0;
                if (h->emb_ok) {
                    const int c = conf_note_voice_burst(h, slot, cc);
                    if (c == CONF_REJECT) {
                        h->emb_ok = 0;
                        conf_reset_slot(h, slot);
                        ended = 1;
                    }
                    open = h->vopen[slot] != 0;
                }
                if (!ended && !open) {
                    if (!h->emb_ok && !is_voice) {
                        conf_reset_slot(h, slot);
                        ended = 1;
                    } else {
                        if (slot == 0) {
                            h->vc1++;
                        } else {
                            h->vc2++;
                        }
                        h->tact_okay = 0;
                        h->emb_ok = 0;
                        action = (h->vc1 > 14 || h->vc2 > 14) ? 2 : 1;
                    }
                } else if (!ended) { /* the voice burst proper */
                    h->skipcount = 0;
                    const int vc = slot == 0 ? h->vc1 : h->vc2;
                    ev_push(ev, pos, ORC_HEV_DMR_CC_PRINT, h->dmr_color_code, vc, slot);
                    if (slot == 0) {
                        h->vc1++;
                    } else {
                        h->vc2++;
                    }
                    h->tact_okay = 0;
                    h->emb_ok = 0;
                    action = (h->vc1 > 14 || h->vc2 > 14) ? 2 : 1;
                }
            }
            if (ended) {
                action = 2;
            }
            ev_push(ev, pos, ORC_HEV_DMR_VOICE_BURST, slot, cc, (is_voice ? 1 : 0) | (action << 4) | (vc_read << 8));
        }
    }
    if (action == 2) {
        return 0;
    }
    /* run_dmr_bs_post_skip() */
    if (h->skipcount > 3) {
        h->tact_okay = 1;
        h->emb_ok = 1;
        return 0;
    }
    return 1;
}

/* one in-frame symbol (stored dibit + reliability).  Returns 1 = reads on, 0 = returned. */
int
orc_dmrh_symbol(orc_dmrh* h, long pos, int d, int rel, orc_hevents* ev) {
    switch (h->phase) {
        case DMRH_FIXED:
        case DMRH_SKIP66:
            if (--h->left <= 0) {
                h->phase = DMRH_IDLE;
                return 0;
            }
            return 1;
        case DMRH_DATA_SUFFIX: {
            h->pay[h->idx] = (uint8_t)d;
            h->rel[h->idx] = (uint8_t)rel;
            if (++h->idx < 95) {
                return 1;
            }
            int burst = 0;
            if (!data_slot_type(h, pos, ev, &burst)) {
                h->phase = DMRH_SKIP66;
                h->left = 66;
                return 1;
            }
            h->phase = DMRH_DATA_SECOND;
            return 1;
        }
        case DMRH_DATA_SECOND:
            h->pay[h->idx] = (uint8_t)d;
            h->rel[h->idx] = (uint8_t)rel;
            if (++h->idx < 144) {
                return 1;
            }
            data_dispatch(h, pos, ev);
            h->phase = DMRH_SKIP66;
            h->left = 66;
            return 1;
        case DMRH_BOOT54:
            h->pay[h->idx] = (uint8_t)d;
            h->rel[h->idx] = (uint8_t)rel;
            if (++h->idx < 144) {
                return 1;
            }
            /* process_dmr_bs_bootstrap_voice_if_open() */
            if (h->vopen[h->boot_slot]) {
                ev_push(ev, pos, ORC_HEV_DMR_CC_PRINT, h->dmr_color_code, 1, h->boot_slot);
            }
            /* dmrBS(): init_dmr_bs_ctx() */
            h->vc1 = h->vc2 = 7;
            if (h->currentslot == 0) {
                h->vc1 = 2;
            } else if (h->currentslot == 1) {
                h->vc2 = 2;
            }
            h->skipcount = 0;
            h->tact_okay = 0;
            h->emb_ok = 0;
            h->internalslot = 0;
            memset(h->red_b, 0, sizeof(h->red_b));
            h->emb_err[0] = h->emb_err[1] = 0;
            h->phase = DMRH_BURST;
            h->idx = 0;
            return 1;
        case DMRH_BURST: {
            h->pay[h->idx] = (uint8_t)d;
            h->rel[h->idx] = (uint8_t)rel;
            const int i = ++h->idx;
            if (i == 12) { /* collect_dmr_bs_cach_and_tact() */
                uint8_t tact[7], cach[24];
                h->tact_okay = tact_decode(h->pay, tact, cach);
                if (!h->tact_okay) {
                    bs_finalize(h, pos, ev);
                    return 0;
                }
                h->internalslot = tact[1];
                h->currentslot = tact[1];
                return 1;
            }
            if (i == 48) { /* is_dmr_bs_redundant_carrier() */
                static const uint8_t at[8] = {16, 27, 1, 32, 3, 33, 13, 7};
                int same = 1;
                for (int k = 0; k < 8; k++) {
                    if (h->pay[12 + at[k]] != h->red_b[at[k]]) {
                        same = 0;
                        break;
                    }
                }
                if (same) {
                    bs_finalize(h, pos, ev);
                    return 0;
                }
                memcpy(h->red_b, h->pay + 12, 36);
                return 1;
            }
            if (i < 144) {
                return 1;
            }
            h->idx = 0;
            if (!bs_burst_done(h, pos, ev)) {
                bs_finalize(h, pos, ev);
                return 0;
            }
            return 1;
        }
        default: return 0;
    }
}

/* a handler that is not restated (MS / direct mode / RC types): the configured symbol count */
int
orc_dmrh_begin_fixed(orc_dmrh* h, int symbols) {
    if (symbols <= 0) {
        h->phase = DMRH_IDLE;
        return 0;
    }
    h->phase = DMRH_FIXED;
    h->left = symbols;
    return 1;
}
