"""Synthetic P25 Phase 1 TSDU (single-block trunking) frames for the end-to-end tests: frame sync + NID
(BCH(63,16,11) + parity) + one 1/2-rate-trellis coded block + status symbols, FM-modulated to cu8 IQ.

Frame layout (TIA-102.BAAA): 180 dibits = 24 FS + 32 NID + 98 block + 21 nulls + 5 status symbols (one after every
35 dibits).  The reference reads it as: NID = 33 dibits after the sync with the status symbol at index 11 dropped
(src/protocol/p25/phase1/dispatch_p25p1.c:123-143); block dibits skip the status positions.
"""
import numpy as np

import fecgen
import orc

FRAME = 180
DUID_TSBK = 7
_LEVEL = np.array([1.0, 3.0, -1.0, -3.0])  # dibit 0,1,2,3


def status_positions():
    return [35, 71, 107, 143, 179]


def block_positions():
    """Frame dibit indices of the 98 coded dibits."""
    pos, i = [], 57
    while len(pos) < 98:
        if i not in status_positions():
            pos.append(i)
        i += 1
    return pos


def crc16_ccitt(bytes10):
    """CRC the reference checks on TSBKs (src/protocol/p25/p25_crc.c:18-36): 0x1021, zero start, inverted."""
    crc = 0
    for v in bytes10:
        for j in range(7, -1, -1):
            bit = (int(v) >> j) & 1
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if ((crc >> 15) & 1) ^ bit else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


def nid_dibits(nac, duid):
    """32 NID dibits: BCH(63,16,11) code word of NAC + DUID, then the parity bit (1 for LDU1 / LDU2, TIA-102.BAAA table 8-4)"""
    data16 = [(nac >> (11 - k)) & 1 for k in range(12)] + [(duid >> (3 - k)) & 1 for k in range(4)]
    cw = list(fecgen.bch_63_16_encode(data16)) + [1 if duid in (0x5, 0xA) else 0]
    return [(cw[2 * k] << 1) | cw[2 * k + 1] for k in range(32)]


def frame_len(blocks):
    """TSDU length in dibits: 24 FS + 32 NID + 98 per block, a status symbol after every 35, padded to a status boundary"""
    return -(-(56 + 98 * blocks) // 35) * 36


def make_frames(rng, n_frames, nac, crc=False, blocks=1):
    """-> (dibits int8 [n_frames * frame_len(blocks)], states [n_frames * blocks, 49]).  With crc=True every block is a
    well-formed TSBK: ten random bytes + their CRC16 (the last-block flag - bit 7 of byte 0, p25p1_tsbk.c:1065 - set on the
    frame's last block only), two bits per trellis state, flushed with a zero state."""
    t = fecgen.tables()
    il = t["il"].astype(np.int64)
    nb = n_frames * blocks
    st = rng.integers(0, 4, (nb, 49)).astype(np.int64)
    if crc:
        for f in range(nb):
            pay = rng.integers(0, 256, 10)
            pay[0] = (int(pay[0]) & 0x7F) | (0x80 if (f % blocks) == blocks - 1 else 0)
            c = crc16_ccitt(pay)
            bits = np.unpackbits(np.array(list(pay) + [c >> 8, c & 0xFF], np.uint8)).astype(np.int64)
            st[f, :48] = (bits[0::2] << 1) | bits[1::2]
            st[f, 48] = 0
    prev = np.concatenate([np.zeros((nb, 1), np.int64), st[:, :-1]], axis=1)
    nib = t["half"][(prev << 2) | st].astype(np.int64)
    dei = np.stack([(nib >> 2) & 3, nib & 3], axis=2).reshape(nb, 98)   # deinterleaved dibits
    tx = dei[:, il]                                                     # transmit order
    nid = nid_dibits(nac, DUID_TSBK)
    flen = frame_len(blocks)
    out = np.zeros((n_frames, flen), np.int8)
    stat = list(range(35, flen, 36))
    pay_pos = [p for p in range(24, flen) if p not in stat]
    for f in range(n_frames):
        fr = np.zeros(flen, np.int8)
        fr[:24] = orc.P25_FS_DIBITS
        fr[pay_pos[:32]] = nid
        fr[pay_pos[32:32 + 98 * blocks]] = tx[f * blocks:(f + 1) * blocks].reshape(-1)
        fr[stat] = 2
        out[f] = fr
    return out.reshape(-1), st


def encode_half_rate(bytes12):
    """12 bytes -> 98 transmit-order dibits of the P25 half-rate trellis (two bits per state, one flush state)"""
    t = fecgen.tables()
    il = t["il"].astype(np.int64)
    bits = np.unpackbits(np.asarray(bytes12, np.uint8)).astype(np.int64)
    st = np.zeros(49, np.int64)
    st[:48] = (bits[0::2] << 1) | bits[1::2]
    prev = np.concatenate([[0], st[:-1]])
    nib = t["half"][(prev << 2) | st].astype(np.int64)
    dei = np.stack([(nib >> 2) & 3, nib & 3], axis=1).reshape(98)
    return dei[il]


def make_pdu(rng, nac, blks, sap=0, good_crc=True):
    """FS + NID (DUID 0xC) + header block (byte 6 = blocks to follow, CRC16 over bytes 0..9) + blks random data blocks, status
    symbols every 36th dibit, padded to a status boundary (p25p1_mdpu.c reads blks + 1 repetitions of 98 data dibits)"""
    hdr = rng.integers(0, 256, 10)
    hdr[1] = (int(hdr[1]) & 0xC0) | (sap & 0x3F)
    hdr[6] = (int(hdr[6]) & 0x80) | (blks & 0x7F)
    c = crc16_ccitt(hdr) ^ (0 if good_crc else 0x5A5A)
    pay = list(encode_half_rate(list(hdr) + [c >> 8, c & 0xFF]))
    for _ in range(blks):
        pay += list(rng.integers(0, 4, 98))
    n_pay = 24 + 32 + len(pay)
    flen = -(-n_pay // 35) * 36
    fr = np.zeros(flen, np.int8)
    stat = list(range(35, flen, 36))
    pos = [p for p in range(flen) if p not in stat]
    body = list(orc.P25_FS_DIBITS) + nid_dibits(nac, 0xC) + pay
    fr[pos[:len(body)]] = body
    fr[stat] = 2
    return fr


def crc32mbf(data, nbits):
    """crc32mbf() of src/protocol/p25/phase1/p25p1_mdpu.c:47-60 over the first nbits of `data` (bytes, MSB first)"""
    crc = 0
    for i in range(nbits):
        crc <<= 1
        b = (int(data[i // 8]) >> (7 - (i % 8))) & 1
        if ((crc >> 32) ^ b) & 1:
            crc ^= 0x04C11DB7
    return (crc & 0xFFFFFFFF) ^ 0xFFFFFFFF


_R34 = None


def _r34_tables():
    global _R34
    if _R34 is None:
        import ctypes as C
        o = orc.oracle()
        for f in ("orc_tbl_r34_point_to_nibble", "orc_tbl_r34_fsm"):
            getattr(o, f).restype = C.POINTER(C.c_uint8)
        il = np.zeros(98, np.uint8)
        o.orc_trellis_interleave_98.argtypes = [C.c_void_p]
        o.orc_trellis_interleave_98(il.ctypes.data)
        _R34 = (np.array(o.orc_tbl_r34_point_to_nibble()[:16]), np.array(o.orc_tbl_r34_fsm()[:64]), il)
    return _R34


def encode_three_quarter_rate(bytes18):
    """18 bytes -> 98 dibits on the air (48 tribits + a flushing zero through the rate 3/4 FSM, interleaved)"""
    p2n, fsm, il = _r34_tables()
    bits = np.unpackbits(np.asarray(bytes18, np.uint8))
    tri = [int(bits[3 * k] << 2 | bits[3 * k + 1] << 1 | bits[3 * k + 2]) for k in range(48)] + [0]
    st, dib = 0, []
    for t in tri:
        nib = int(p2n[fsm[st * 8 + t] & 15])
        dib += [nib >> 2, nib & 3]
        st = t
    return np.array(dib, np.uint8)[il]            # received dibit i carries de-interleaved dibit il[i]


def crc9(bits135):
    """ComputeCrc9Bit (src/protocol/dmr/dmr_utils.c:410-435): polynomial 0x059 over the bits, inverted"""
    crc = 0
    for b in bits135:
        crc = ((crc << 1) ^ 0x059) if (((crc >> 8) & 1) ^ int(b)) else (crc << 1)
    return (crc & 0x1FF) ^ 0x1FF


def confirmed_block(dbsn, payload16, good_crc9=True):
    """DBSN(7) | CRC9(9) | 16 payload bytes -> 18 bytes"""
    bits = [(dbsn >> (6 - i)) & 1 for i in range(7)] + list(np.unpackbits(np.asarray(payload16, np.uint8)))
    c = crc9(bits) ^ (0 if good_crc9 else 0x011)
    return np.array([((dbsn & 0x7F) << 1) | (c >> 8), c & 0xFF] + [int(x) for x in payload16], np.uint8)


def make_pdu_coded(rng, nac, blks, sap=0, good_crc16=True, good_crc32=True, confirmed=False, header_reps=0, bad_crc9_at=(), combined=False):
    """a data unit as the reference decodes it: header block (CRC16), blks half-rate coded data blocks whose last four bytes are the
    CRC32 of the rest -> (dibits, header12, data [blks][12]); confirmed = True: A/N = 1, format 0x16 and rate 3/4 blocks
    (data [blks][18]).  header_reps > 0 (with good_crc16 = False and blks = 0 in
    the header): the first header fails its CRC16 and header_reps good copies follow, as the reference's repetition fallback expects.
    combined = True (blks = 0): three copies of a good header, each with a different third of its 98 dibits sent with the wrong sign -
    none decodes on its own, the position-wise sum of their LLRs does (p25_mpdu_try_combined_header)"""
    hdr = rng.integers(0, 256, 10)
    hdr[0] = (int(hdr[0]) & 0xA0) | ((0x40 | 0x16) if confirmed else (int(hdr[0]) & 0x0F))      # AN / format
    hdr[1] = (int(hdr[1]) & 0xC0) | (sap & 0x3F)
    hdr[6] = (int(hdr[6]) & 0x80) | (blks & 0x7F)
    c = crc16_ccitt(hdr)
    good = np.array(list(hdr) + [c >> 8, c & 0xFF], np.uint8)
    bad = good.copy()
    bad[10] ^= 0x5A
    bad[3] ^= 0x81
    first = good if good_crc16 else bad
    pay = list(encode_half_rate(list(first)))
    data = np.zeros((max(blks, header_reps), 12), np.uint8)
    if combined:
        pay = []
        for k, (a, b) in enumerate(((0, 33), (33, 66), (66, 98))):
            d = np.array(encode_half_rate(list(good)))
            d[a:b] ^= 2
            pay += list(d)
        data = np.stack([good, good])
    elif header_reps:
        for k in range(header_reps):
            data[k] = good
    elif blks and confirmed and good_crc16:       # confirmed data: rate 3/4 blocks of DBSN | CRC9 | 16 bytes, CRC32 ends the payload
        flat = rng.integers(0, 256, 16 * blks).astype(np.uint8)
        c32 = crc32mbf(flat, 128 * blks - 32) ^ (0 if good_crc32 else 0x00010000)
        flat[-4:] = [(c32 >> 24) & 0xFF, (c32 >> 16) & 0xFF, (c32 >> 8) & 0xFF, c32 & 0xFF]
        data = np.stack([confirmed_block(k, flat[16 * k:16 * k + 16], k not in bad_crc9_at) for k in range(blks)])
        for k in range(blks):
            pay += list(encode_three_quarter_rate(data[k]))
    elif blks:
        flat = rng.integers(0, 256, 12 * blks).astype(np.uint8)
        c32 = crc32mbf(flat, 96 * blks - 32) ^ (0 if good_crc32 else 0x00010000)
        flat[-4:] = [(c32 >> 24) & 0xFF, (c32 >> 16) & 0xFF, (c32 >> 8) & 0xFF, c32 & 0xFF]
        data = flat.reshape(blks, 12)
    if data.shape[1] == 12 and not combined:
        for k in range(len(data)):
            pay += list(encode_half_rate(list(data[k])))
    n_pay = 24 + 32 + len(pay)
    flen = -(-n_pay // 35) * 36
    fr = np.zeros(flen, np.int8)
    stat = list(range(35, flen, 36))
    pos = [p for p in range(flen) if p not in stat]
    body = list(orc.P25_FS_DIBITS) + nid_dibits(nac, 0xC) + pay
    fr[pos[:len(body)]] = body
    fr[stat] = 2
    return fr, first, data


def frame_with_duid(rng, nac, duid, n_body):
    """FS + NID of the given DUID + n_body random dibits (status symbols 2 where the frame has them)"""
    fr = np.zeros(24 + 33 + n_body, np.int8)
    fr[:24] = orc.P25_FS_DIBITS
    fr[24:] = rng.integers(0, 4, 33 + n_body)
    nid_pos = [p for p in range(24, 57) if p != 35]
    fr[nid_pos] = nid_dibits(nac, duid)
    fr[35::36] = 2
    return fr


def _walk(idx, n):
    """the next n frame positions from idx on that are not status symbols -> (positions, next idx)"""
    out = []
    while len(out) < n:
        if idx % 36 != 35:
            out.append(idx)
        idx += 1
    return out, idx


def make_hdu(rng, nac):
    """One header data unit (p25p1_hdu.c:191-268): 20 random hex words -> RS(36,20,17) -> Golay(24,6) per word, hex_data[19..0] then
    hex_parity[15..0] on the air -> (dibits int8 [396], hex words sent [20])"""
    d = rng.integers(0, 64, 20)
    syms = np.concatenate([d, fecgen.rs63_encode(d, 8)])
    fr = np.full(396, 2, np.int8)
    fr[:24] = orc.P25_FS_DIBITS
    fr[[q for q in range(24, 57) if q != 35]] = nid_dibits(nac, 0x0)
    idx = 57
    for seq in range(36):
        word = 19 - seq if seq < 20 else 20 + (15 - (seq - 20))
        hp, idx = _walk(idx, 3)
        pp, idx = _walk(idx, 6)
        b6 = np.array([(int(syms[word]) >> (5 - k)) & 1 for k in range(6)], np.uint8)
        d12 = np.zeros(12, np.uint8)
        d12[6:] = b6
        par = np.array(fecgen.golay24_encode(d12), np.uint8)
        fr[hp] = (b6[0::2] << 1) | b6[1::2]
        fr[pp] = (par[0::2] << 1) | par[1::2]
    return fr, d


def make_tdulc(rng, nac):
    """One terminator with link control (p25p1_tdulc.c:199-207): 12 random hex words -> RS(24,12,13) -> six + six Golay(24,12) words
    -> (dibits int8 [216], hex words sent [12])"""
    d = rng.integers(0, 64, 12)
    hexw = np.concatenate([d, fecgen.rs63_encode(d, 6)])
    fr = np.full(216, 2, np.int8)
    fr[:24] = orc.P25_FS_DIBITS
    fr[[q for q in range(24, 57) if q != 35]] = nid_dibits(nac, 0xF)
    idx = 57
    for seq in range(12):
        w = 5 - seq if seq < 6 else 6 + (5 - (seq - 6))
        dp, idx = _walk(idx, 6)
        pp, idx = _walk(idx, 6)
        hi, lo = int(hexw[2 * w + 1]), int(hexw[2 * w])
        b12 = np.array([(hi >> (5 - k)) & 1 for k in range(6)] + [(lo >> (5 - k)) & 1 for k in range(6)], np.uint8)
        par = np.array(fecgen.golay24_encode(b12), np.uint8)
        fr[dp] = (b12[0::2] << 1) | b12[1::2]
        fr[pp] = (par[0::2] << 1) | par[1::2]
    return fr, d


def make_tdu(nac):
    """Simple terminator: FS + NID (DUID 3) + 14 null dibits and the status symbol -> int8 [72]"""
    fr = np.zeros(72, np.int8)
    fr[:24] = orc.P25_FS_DIBITS
    fr[[q for q in range(24, 57) if q != 35]] = nid_dibits(nac, 0x3)
    fr[35::36] = 2
    return fr


def weaken_nid(dibits, frame_start, rng, strong=10, weak=3):
    """damage the NID of the frame at dibit `frame_start` so that the BCH hard decode fails (strong + weak > 11 bit errors)
    and the Chase search over the least reliable bits repairs it: `strong` symbols get the opposite sign (one bit error
    each, full reliability), `weak` outer symbols are pulled just inside the inner threshold (one low-reliability bit error
    each).  Returns the per-symbol amplitude scale for modulate_disc()."""
    scale = np.ones(len(dibits))
    pos = [frame_start + p for p in range(24, 57) if p != 35]
    pick = rng.permutation(len(pos))
    n_s = 0
    for k in pick:
        if n_s >= strong:
            break
        dibits[pos[k]] ^= 2          # sign flip: high bit wrong, magnitude kept
        n_s += 1
    n_w = 0
    for k in pick[strong:]:
        if n_w >= weak:
            break
        if dibits[pos[k]] in (1, 3):  # outer level: pull it to 0.55 of full scale -> sliced as the inner level, low bit wrong
            scale[pos[k]] = 0.55
            n_w += 1
    return scale


def modulate_disc(dibits, lead=300, noise=100.0, seed=0, sps=10, amp=7000.0, tail=400, scale=None):
    """Dibit stream -> discriminator-scale float32 samples (smoothed 4-level), `lead` noise-only samples first; scale = optional
    per-symbol amplitude factors"""
    rng = np.random.default_rng(seed)
    win = np.hanning(sps + 3)[1:-1]
    win /= win.sum()
    lv = _LEVEL[dibits] * (1.0 if scale is None else scale)
    shaped = np.convolve(np.repeat(lv, sps), win, mode="same") * amp
    x = np.concatenate([np.zeros(lead), shaped, np.zeros(tail)])
    return (x + rng.normal(0.0, noise, x.shape)).astype(np.float32)


def modulate_cu8(dibits, n, sps=10, dev=0.06, lead=230, seed=0, noise=0.02):
    """Dibit stream -> uint8 [n, 2] C4FM (smoothed 4-level FM), `lead` idle samples first."""
    rng = np.random.default_rng(seed)
    lv = _LEVEL[dibits]
    nrz = np.repeat(lv, sps)
    win = np.hanning(sps + 3)[1:-1]
    win /= win.sum()
    shaped = np.convolve(nrz, win, mode="same")
    f = np.zeros(n)
    m = min(n - lead, len(shaped))
    f[lead:lead + m] = shaped[:m]
    ph = 0.3 + np.cumsum(f * dev)
    i = 0.8 * np.cos(ph) + rng.normal(0, noise, n)
    q = 0.8 * np.sin(ph) + rng.normal(0, noise, n)
    out = np.empty((n, 2), np.uint8)
    out[:, 0] = np.clip(np.rint(127.5 + 127.5 * i), 0, 255)
    out[:, 1] = np.clip(np.rint(127.5 + 127.5 * q), 0, 255)
    return out


def extract_frames(rec4, flags, count):
    """Receive-loop output of one channel -> per accepted sync: (nid bits63, nid rel63, parity, parity_rel, llr196)."""
    acc = np.flatnonzero(flags[:count] & 2)
    bp = [p - 24 for p in block_positions()]
    nid_idx = [k for k in range(33) if k != 11]
    frames = []
    for a in acc:
        if a + 1 + (FRAME - 24) > count:
            break
        d = rec4[a + 1:a + 1 + FRAME - 24]
        nd = d[nid_idx]
        bits = np.stack([(nd[:, 0] >> 1) & 1, nd[:, 0] & 1], axis=1).reshape(64).astype(np.uint8)
        # per-bit reliability = min(|llr of that bit|, 255) (dispatch_p25p1.c:59-83,138-142), not the dibit's byte
        rel = np.minimum(np.abs(np.stack([nd[:, 2], nd[:, 3]], axis=1)), 255).reshape(64).astype(np.uint8)
        blk = d[bp]
        llr = np.stack([blk[:, 2], blk[:, 3]], axis=1).reshape(196).astype(np.int16)
        frames.append((bits[:63], rel[:63], int(bits[63]), int(rel[63]), llr))
    return acc, frames


def expected_half_rate_output(states):
    """What the 1/2-rate decoder returns for an error-free block: run the oracle on ideal LLRs of the same states."""
    t = fecgen.tables()
    n = states.shape[0]
    prev = np.concatenate([np.zeros((n, 1), np.int64), states[:, :-1]], axis=1)
    nib = t["half"][(prev << 2) | states].astype(np.int64)
    bits = np.stack([(nib >> 3) & 1, (nib >> 2) & 1, (nib >> 1) & 1, nib & 1], axis=2).reshape(n, 196)
    llr_dei = ((2 * bits - 1) * 200).astype(np.int16)
    il = t["il"].astype(np.int64)
    rx = np.zeros((n, 196), np.int16)
    rx[:, 0::2] = llr_dei[:, 2 * il]
    rx[:, 1::2] = llr_dei[:, 2 * il + 1]
    out, _ = fecgen.oracle_p25_half_rate(np.ascontiguousarray(rx))
    return out


# ---- LDU1 field map (reading order of src/protocol/p25/phase1/p25p1_ldu1.c:185-216; status dibit after every 35) ----
def ldu1_positions():
    """Frame dibit indices (0 = first dibit of the frame sync) of the 24 Hamming(10,6,3) words of an LDU1:
    returns (data_pos[12][5], parity_pos[12][5]) with word index as in hex_data[w] / hex_parity[w]; each word is
    3 data dibits then 2 parity dibits, dibit bit 1 first."""
    idx = 57                                   # 24 FS + 33 NID (incl. its status dibit)

    def take(n):
        nonlocal idx
        out = []
        while len(out) < n:
            if idx % 36 == 35:
                idx += 1                       # status symbol
                continue
            out.append(idx)
            idx += 1
        return out

    data = [None] * 12
    par = [None] * 12
    take(72)
    take(72)
    for w in (11, 10, 9, 8):
        data[w] = take(5)
    take(72)
    for w in (7, 6, 5, 4):
        data[w] = take(5)
    take(72)
    for w in (3, 2, 1, 0):
        data[w] = take(5)
    take(72)
    for w in (11, 10, 9, 8):
        par[w] = take(5)
    take(72)
    for w in (7, 6, 5, 4):
        par[w] = take(5)
    take(72)
    for w in (3, 2, 1, 0):
        par[w] = take(5)
    take(72)
    take(16)
    take(72)
    assert idx in (863, 864)
    return np.array(data), np.array(par)


def ldu2_positions():
    """LDU2 (src/protocol/p25/phase1/p25p1_ldu2.c:211-236): same slot map as LDU1, 16 data words (hex_data[15..0]) in
    the first four slots and 8 parity words (hex_parity[7..0]) in the next two."""
    d1, p1 = ldu1_positions()
    slots = [d1[[11, 10, 9, 8]], d1[[7, 6, 5, 4]], d1[[3, 2, 1, 0]], p1[[11, 10, 9, 8]], p1[[7, 6, 5, 4]], p1[[3, 2, 1, 0]]]
    data = [None] * 16
    par = [None] * 8
    for s in range(4):
        for k in range(4):
            data[15 - 4 * s - k] = slots[s][k]
    for s in range(2):
        for k in range(4):
            par[7 - 4 * s - k] = slots[4 + s][k]
    return np.array(data), np.array(par)


# ---- LDU1 / LDU2 voice traffic (encoder side; reading order as ldu1_positions above / ddn_host_p25_layout.c) ----------
DUID_LDU1, DUID_LDU2 = 0x5, 0xA
LDU = 864
_NID_PARITY = {0x5: 1, 0xA: 1}       # k_duid_parity_table, src/protocol/p25/phase1/p25p1_check_nid.cpp:44-49
_CACHE = {}


def _hamming_10_6_3_parity():
    """6 data bits -> 4 parity bits of the P25 Hamming(10,6,3) word: the one parity the oracle's decoder accepts with no
    correction (src/fec/hamming_10_6_3.cpp restated in oracle/ddn_oracle_block.c)."""
    if "ham" not in _CACHE:
        import ctypes as C
        o = orc.oracle()
        o.orc_hamming_10_6_3.argtypes = [C.c_int, C.c_void_p]
        tab = np.zeros(64, np.int64)
        for d in range(64):
            for p in range(16):
                f = C.c_int(0)
                if o.orc_hamming_10_6_3((d << 4) | p, C.byref(f)) == 0:
                    tab[d] = p
                    break
            else:
                raise AssertionError(d)
        _CACHE["ham"] = tab
    return _CACHE["ham"]


def _imbe_interleave_map():
    """(row, col) of the IMBE code vector array carried by bit b (0 = high bit of dibit 0) of a voice frame's 72 dibits:
    the inverse of process_IMBE's schedule, read off the oracle's de-interleaver with status skipping out of the way."""
    if "imbe" not in _CACHE:
        m = []
        for b in range(144):
            d = np.zeros(80, np.uint8)
            d[b // 2] = 2 if b % 2 == 0 else 1
            fr, _, _, _, _ = orc.oracle_imbe_deinterleave(d, np.zeros(80, np.int16), np.zeros(80, np.int16), 36 + 40)
            rc = np.argwhere(fr)
            assert rc.shape == (1, 2), (b, rc)
            m.append((int(rc[0, 0]), int(rc[0, 1])))
        assert len(set(m)) == 144
        _CACHE["imbe"] = m
    return _CACHE["imbe"]


def imbe_frame_to_dibits(fr):
    """u8 [8][23] code vectors (mbelib layout) -> the 72 dibits that carry them on the air."""
    bits = np.array([fr[r, c] for (r, c) in _imbe_interleave_map()], np.int64)
    return ((bits[0::2] << 1) | bits[1::2]).astype(np.int8)


def _lsd_codeword(d8):
    import ctypes as C
    o = orc.oracle()
    p = o.orc_p25_lsd_parity(int(d8))
    bits = [(d8 >> (7 - k)) & 1 for k in range(8)] + [(p >> (7 - k)) & 1 for k in range(8)]
    return [(bits[2 * k] << 1) | bits[2 * k + 1] for k in range(8)]


def make_ldus(rng, n_ldus, nac, imbe_frames):
    """-> (dibits int8 [n_ldus * 864], hex words) for alternating LDU1 / LDU2 frames carrying imbe_frames
    u8 [n_ldus * 9][8][23] (already FEC-encoded code vectors, tests/mbe.py:imbe_encode)."""
    ham = _hamming_10_6_3_parity()
    out = np.zeros((n_ldus, LDU), np.int8)
    words_sent = []
    for f in range(n_ldus):
        ldu = 1 + (f & 1)
        duid = DUID_LDU1 if ldu == 1 else DUID_LDU2
        data16 = [(nac >> (11 - k)) & 1 for k in range(12)] + [(duid >> (3 - k)) & 1 for k in range(4)]
        cw = list(fecgen.bch_63_16_encode(data16)) + [_NID_PARITY[duid]]
        nid = [(cw[2 * k] << 1) | cw[2 * k + 1] for k in range(32)]
        n_data, t = (12, 6) if ldu == 1 else (16, 4)
        d = rng.integers(0, 64, n_data)
        syms = list(d) + list(fecgen.rs63_encode(d, t))          # hex_data[0..], hex_parity[0..]
        words_sent.append(np.array(syms))
        word_dibits = []
        for s in syms:
            b10 = [(int(s) >> (5 - k)) & 1 for k in range(6)] + [(int(ham[int(s)]) >> (3 - k)) & 1 for k in range(4)]
            word_dibits.append([(b10[2 * k] << 1) | b10[2 * k + 1] for k in range(5)])
        dpos, ppos = ldu1_positions() if ldu == 1 else ldu2_positions()
        fr = np.zeros(LDU, np.int8)
        fr[:24] = orc.P25_FS_DIBITS
        fr[[p for p in range(24, 57) if p != 35]] = nid
        for w in range(n_data):
            fr[dpos[w]] = word_dibits[w]
        for w in range(24 - n_data):
            fr[ppos[w]] = word_dibits[n_data + w]
        import ddn
        import ctypes as C
        first9 = np.zeros(9, np.int32)
        st9 = np.zeros(9, np.int32)
        ddn.lib().ddn_p25p1_layout_ldu_imbe(first9.ctypes.data, st9.ctypes.data)
        for v in range(9):
            idx, got = int(first9[v]), 0
            vd = imbe_frame_to_dibits(imbe_frames[f * 9 + v])
            while got < 72:
                if idx % 36 == 35:
                    idx += 1
                    continue
                fr[idx] = vd[got]
                got += 1
                idx += 1
        lsd = np.zeros(16, np.int32)
        ddn.lib().ddn_p25p1_layout_ldu_lsd(lsd.ctypes.data)
        fr[lsd] = _lsd_codeword(int(rng.integers(0, 256))) + _lsd_codeword(int(rng.integers(0, 256)))
        fr[35::36] = 2                                               # status symbols
        out[f] = fr
    return out.reshape(-1), words_sent
