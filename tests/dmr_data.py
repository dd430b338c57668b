"""The DMR chain's data-burst and embedded-link-control stages restated over the CPU oracle's receive-loop output (records + handler
events of a whole stream): what dmr_data_burst_handler() computes before it hands over to the protocol layer
(src/protocol/dmr/dmr_dburst.c:323-650) with the oracle's block decoders (oracle/ddn_oracle_fec*.c, each pinned to the compiled
reference by the tests/test_oracle_*.py files)."""
import ctypes as C

import numpy as np

import bptc_small
import dmrgen
import fec3
import fecgen
import orc
import p25gen
import rx4

CRC_LEN = {0: 16, 1: 24, 2: 24, 3: 16, 4: 16, 5: 0, 6: 16, 7: 9, 8: 9, 9: 0, 10: 9, 11: 16}


def _int(bits):
    v = 0
    for b in bits:
        v = (v << 1) | int(b)
    return v


def tables34():
    p2n, fsm, il = p25gen._r34_tables()
    return p2n, fsm, il


def candidate_metric(td98, rel98, b18):
    """dmr_r34_candidate_metric() (src/protocol/dmr/dmr_34_viterbi.c:410-444)"""
    p2n, fsm, il = tables34()
    dei, rdei = np.zeros(98, np.int64), np.zeros(98, np.int64)
    dei[il] = np.asarray(td98, np.int64) & 3
    rdei[il] = np.asarray(rel98, np.int64)
    bits = np.unpackbits(np.asarray(b18, np.uint8))
    states = [_int(bits[3 * k:3 * k + 3]) for k in range(48)] + [0]
    metric, prev = 0, 0
    for t in range(49):
        x = int(p2n[fsm[prev * 8 + states[t]] & 15]) ^ int((dei[2 * t] << 2) | dei[2 * t + 1])
        metric += (((x >> 3) & 1) + ((x >> 2) & 1)) * int(rdei[2 * t]) + (((x >> 1) & 1) + (x & 1)) * int(rdei[2 * t + 1])
        prev = states[t]
    return metric


def r34_pool(td98, rel98):
    """dmr_dburst_pick_trellis_payload() (:502-536) -> (pool [(bytes18, metric, crc_ok, dbsn)], unconfirmed pick, confirmed pick before
    the DBSN expectation, whether a candidate with a good CRC9 exists)"""
    o = fecgen.bind_oracle_fec()
    td = np.ascontiguousarray(td98, np.uint8)
    rel = np.ascontiguousarray(rel98, np.uint8)
    hard = fecgen.oracle_r34(td[None], None)[0]
    soft = fecgen.oracle_r34(td[None], rel[None])[0]
    o.orc_r34_decode_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lm, lb = np.zeros(32, np.int32), np.zeros((32, 18), np.uint8)
    nl = o.orc_r34_decode_list(td.ctypes.data, rel.ctypes.data, 32, lm.ctypes.data, lb.ctypes.data)
    pool = []
    n_hs = 0
    for k, b in enumerate([hard, soft] + [lb[j] for j in range(nl)]):
        if not any(np.array_equal(b, q[0]) for q in pool) and len(pool) < 34:
            bits = np.unpackbits(b)
            c9 = p25gen.crc9(list(bits[16:144]) + list(bits[:7]))
            pool.append((b.copy(), candidate_metric(td, rel, b), int(c9 == (_int(bits[7:16]) ^ 0x1FF)), int(b[0]) >> 1))
        if k == 1:
            n_hs = len(pool)

    def lowest(idx):
        best = -1
        for j in idx:
            if best < 0 or pool[j][1] < pool[best][1]:
                best = j
        return best
    u = lowest(range(n_hs))
    crc = lowest([j for j in range(len(pool)) if pool[j][2]])
    c = crc if crc >= 0 else lowest(range(len(pool)))
    return pool, pool[u][0], pool[c][0], int(crc >= 0)


def data_burst(dib144, rel144):
    """one dispatched burst -> dict(type, bits96, bytes12, errs, crc, info, r34...)"""
    dib = np.asarray(dib144, np.int64) & 3
    st, info, _ = rx4.dmr_burst_fields(dib[:90], dib[90:], 0)
    got, _, ok = fec3.oracle_decode(5, st[None])
    out = dict(info=info.copy(), type=0xFF)
    bits, r3, errs = fec3.oracle_bptc(info[None], 1)
    out["bits96"], out["errs"] = bits[0], int(errs[0])
    out["undefined"] = bool(fec3.oracle_bptc.undefined[0])
    by = np.packbits(bits[0])
    out["bytes12"] = by.copy()
    out["crc"] = 0
    if not ok[0]:
        return out
    ty = _int(got[0][4:8])
    out["type"] = ty
    flags = 0
    b = bits[0]
    if ty in (1, 2):
        m = dmrgen.CRC_MASK[ty]
        cw = by.copy()
        cw[9] ^= (m >> 16) & 0xFF
        cw[10] ^= (m >> 8) & 0xFF
        cw[11] ^= m & 0xFF
        o = orc.oracle()
        o.orc_rs_12_9.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        syn, found = np.zeros(3, np.uint8), C.c_uint8(0)
        res = o.orc_rs_12_9(cw.ctypes.data, syn.ctypes.data, C.byref(found))
        if res in (0, 1):
            cw[9] ^= (m >> 16) & 0xFF
            cw[10] ^= (m >> 8) & 0xFF
            cw[11] ^= m & 0xFF
            out["bytes12"] = cw
            flags |= 1 | (4 if res == 1 else 0)
    elif ty == 7:
        flags |= 1
        if p25gen.crc9(list(b[16:96]) + list(b[:7])) == (_int(b[7:16]) ^ 0x0F0):
            flags |= 2
    elif ty == 8:
        flags |= 1
        td = np.concatenate([dib[12:61], dib[95:144]]).astype(np.uint8)
        rel = np.concatenate([rel144[12:61], rel144[95:144]]).astype(np.uint8)
        out["pool"], out["unconfirmed"], out["confirmed"], out["confirmed_crc"] = r34_pool(td, rel)
    elif ty == 10:
        flags |= 1
        if dmrgen.crc9_confirmed_rate1(info) == _int(info[7:16]):
            flags |= 2
    elif ty <= 11 and ty != 9:
        n = CRC_LEN[ty]
        ext = (_int(b[96 - n:]) if n else 0) ^ dmrgen.CRC_MASK.get(ty, 0)
        if (rx4.crc_ccitt_bits(b[:80]) ^ 0xFFFF) == ext:
            flags |= 1
    out["crc"] = flags
    return out


def stream_expectation(w, events):
    """-> (data bursts [(pos of the last symbol, slot, result dict)] in air order, embedded LCs per slot [(pos, lc77, errs, ok, undefined)])"""
    rec = w["rec4"]
    syncs = {int(p): i for i, p in enumerate(w["sync_pos"])}
    data = []
    sig = np.zeros((2, 7, 48), np.uint8)
    lcs = [[], []]
    for (pos, kind, a, b, c) in events:
        if kind == 6 and b == 0:
            dib = (rec[pos - 143:pos + 1, 0] & 3).astype(np.uint8)
            rel = (rec[pos - 143:pos + 1, 1] & 0xFF).astype(np.uint8)
            if pos - 54 in syncs:         # found by the sync search: the reference reads the first 90 dibits (and their reliabilities)
                i = syncs[pos - 54]      # from its payload / soft history (dmr_data.c:56-100) - not always what the symbol records hold
                dib[:90] = w["pre"][i] & 3
                rel[:90] = w["pre_rel"][i]
            data.append((int(pos), int(c) & 1, data_burst(dib, rel)))
        elif kind == 7:
            vcr = (c >> 8) & 0xFF
            if 1 < vcr < 7:
                d = rec[pos - 143 + 66:pos - 143 + 90, 0] & 3
                sig[a & 1, vcr - 1, 0::2] = (d >> 1) & 1
                sig[a & 1, vcr - 1, 1::2] = d & 1
        elif kind == 6 and b == 6:
            slot = c & 1
            m = np.zeros((8, 16), np.uint8)
            q, burst = 0, 1
            for col in range(16):
                for row in range(8):
                    m[row, col] = sig[slot, burst, q + 8]
                    q += 1
                    if q >= 32:
                        q, burst = 0, burst + 1
            errs, lc, row0 = bptc_small.oracle_128x77(m.reshape(-1))
            by = np.packbits(lc[:72])
            ok = int(int(by.astype(np.int64).sum()) % 31 == _int(lc[72:77]))
            lcs[slot].append((int(pos), lc.copy(), int(errs), ok, bool(row0)))
    return data, lcs
