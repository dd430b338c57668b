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


def make_frames(rng, n_frames, nac, crc=False):
    """-> (dibits int8 [n_frames*180], states [n_frames,49]).  With crc=True every block is a well-formed TSBK: ten random
    bytes + their CRC16, two bits per trellis state, flushed with a zero state."""
    t = fecgen.tables()
    il = t["il"].astype(np.int64)
    st = rng.integers(0, 4, (n_frames, 49)).astype(np.int64)
    if crc:
        for f in range(n_frames):
            pay = rng.integers(0, 256, 10)
            c = crc16_ccitt(pay)
            bits = np.unpackbits(np.array(list(pay) + [c >> 8, c & 0xFF], np.uint8)).astype(np.int64)
            st[f, :48] = (bits[0::2] << 1) | bits[1::2]
            st[f, 48] = 0
    prev = np.concatenate([np.zeros((n_frames, 1), np.int64), st[:, :-1]], axis=1)
    nib = t["half"][(prev << 2) | st].astype(np.int64)
    dei = np.stack([(nib >> 2) & 3, nib & 3], axis=2).reshape(n_frames, 98)   # deinterleaved dibits
    tx = dei[:, il]                                                          # transmit order
    data16 = [(nac >> (11 - k)) & 1 for k in range(12)] + [(DUID_TSBK >> (3 - k)) & 1 for k in range(4)]
    cw = list(fecgen.bch_63_16_encode(data16)) + [0]                          # parity bit 0 for DUID 7
    nid = [(cw[2 * k] << 1) | cw[2 * k + 1] for k in range(32)]
    out = np.zeros((n_frames, FRAME), np.int8)
    bp = block_positions()
    for f in range(n_frames):
        fr = np.zeros(FRAME, np.int8)
        fr[:24] = orc.P25_FS_DIBITS
        nid_pos = [p for p in range(24, 57) if p != 35]
        fr[nid_pos] = nid
        fr[bp] = tx[f]
        fr[status_positions()] = 2
        out[f] = fr
    return out.reshape(-1), st


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
