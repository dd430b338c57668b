"""P25 Phase 2 RS(63,35) sections (ESS / FACCH / SACCH): test-side encoder, traffic generator and the oracle / reference
callers.  Block position p = 0..62 carries the coefficient of x^(62-p); generator roots alpha^1..alpha^28 over GF(64)
(x^6 + x + 1), the code of src/fec/ez.cpp (ezpwd RS<63,35>)."""
import ctypes as C

import numpy as np

import orc

KINDS = {"ess": 0, "facch": 1, "sacch": 2}
N_DATA = (16, 26, 30)
N_PAR = (28, 19, 22)
FIRST = (19, 9, 5)

_EX = np.zeros(126, np.int64)
_LG = np.zeros(64, np.int64)
_x = 1
for _i in range(63):
    _EX[_i] = _EX[_i + 63] = _x
    _LG[_x] = _i
    _x <<= 1
    if _x & 0x40:
        _x ^= 0x43


def gmul(a, b):
    return int(_EX[_LG[a] + _LG[b]]) if a and b else 0


def _generator():
    g = [1]
    for r in range(1, 29):           # times (x + alpha^r), coefficients lowest degree first
        root = int(_EX[r])
        ng = [0] * (len(g) + 1)
        for i, c in enumerate(g):
            ng[i + 1] ^= c
            ng[i] ^= gmul(c, root)
        g = ng
    return g                         # degree 28, monic


_GEN = _generator()


def encode_block(data35):
    """35 data symbols (block positions 0..34) -> 63-symbol block with the 28 parity symbols at 35..62."""
    rem = [0] * 28                   # remainder register, rem[27] = highest degree
    for d in data35:
        fb = int(d) ^ rem[27]
        for i in range(27, 0, -1):
            rem[i] = rem[i - 1] ^ gmul(fb, _GEN[i])
        rem[0] = gmul(fb, _GEN[0])
    return np.array(list(data35) + rem[::-1], np.uint8)


def bits_of(symbols):
    s = np.asarray(symbols, np.uint8)
    return ((s[:, None] >> np.arange(5, -1, -1)[None, :]) & 1).astype(np.int32).reshape(-1)


def make_case(rng, kind, n_err, n_extra_erasures, erase_hits):
    """One received section: random message, n_err symbol errors at transmitted positions, the punctured parity positions
    declared erased (FACCH / SACCH) plus n_extra_erasures more, erase_hits of which land on corrupted symbols.
    -> (payload bits int32, parity bits int32, erasures int32 (the reference's position convention), sent payload bits)"""
    k = KINDS[kind] if isinstance(kind, str) else kind
    nd, npar, first = N_DATA[k], N_PAR[k], FIRST[k]
    data35 = np.zeros(35, np.uint8)
    data35[first:35] = rng.integers(0, 64, nd)
    blk = encode_block(data35)
    sent = blk.copy()
    tx = np.arange(first, 35 + npar)                      # transmitted block positions
    bad = rng.choice(tx, size=min(n_err, tx.size), replace=False) if n_err else np.zeros(0, np.int64)
    for p in bad:
        blk[p] ^= rng.integers(1, 64)
    blk[35 + npar:] = 0                                   # punctured parity: never sent
    er = list(range(35 + npar, 63))
    hits = list(rng.choice(bad, size=min(erase_hits, len(bad)), replace=False)) if erase_hits and len(bad) else []
    rest = [p for p in tx if p not in bad]
    more = list(rng.choice(rest, size=min(max(n_extra_erasures - len(hits), 0), len(rest)), replace=False)) if n_extra_erasures else []
    er = er + [int(p) for p in hits] + [int(p) for p in more]
    rng.shuffle(er)
    er = er[:28]
    if k == 0:
        er = [p - 19 for p in er]                         # ESS positions count from the first payload symbol
    return (bits_of(blk[first:35]), bits_of(blk[35:35 + npar]), np.array(er, np.int32), bits_of(sent[first:35]))


def oracle_rs28(kind, payload, parity, erasures):
    o = orc.oracle()
    o.orc_ez_rs28.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    pl = np.ascontiguousarray(payload, np.int32).copy()
    pa = np.ascontiguousarray(parity, np.int32)
    er = np.ascontiguousarray(erasures, np.int32)
    rc = o.orc_ez_rs28(kind, pl.ctypes.data, pa.ctypes.data, er.ctypes.data if er.size else None, int(er.size))
    return pl, rc


def ref_rs28(kind, payload, parity, erasures):
    r = C.CDLL(orc.REF_SO)
    fn = (r.ez_rs28_ess, r.ez_rs28_facch, r.ez_rs28_sacch)[kind]
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    pl = np.ascontiguousarray(payload, np.int32).copy()
    pa = np.ascontiguousarray(parity, np.int32).copy()
    er = np.ascontiguousarray(erasures, np.int32)
    rc = fn(pl.ctypes.data, pa.ctypes.data, er.ctypes.data if er.size else None, int(er.size))
    return pl, rc


# ---- P25 Phase 2 FACCH / SACCH bursts (p25p2_frame.c:473-495,652-671: where the section's bits sit in a 360-bit timeslot) ----------
XCCH_PAYLOAD_POS = {0: list(range(2, 74)) + list(range(76, 138)) + list(range(180, 202)),
                    1: list(range(2, 74)) + list(range(76, 184))}
XCCH_PARITY_POS = {0: list(range(202, 244)) + list(range(246, 318)), 1: list(range(184, 244)) + list(range(246, 318))}


def make_xcch_burst(rng, kind, n_err, weak_errors, n_weak_good=0, strong=200):
    """kind 0 FACCH / 1 SACCH -> (bits360 u8, llr360 i16, sent payload bits): a valid RS(63,35) section laid into the burst, n_err
    corrupted symbols of which weak_errors carry low |LLR| (what the ranked erasures find), n_weak_good clean symbols made weak too"""
    k = kind + 1
    nd, npar, first = N_DATA[k], N_PAR[k], FIRST[k]
    data35 = np.zeros(35, np.uint8)
    data35[first:35] = rng.integers(0, 64, nd)
    blk = encode_block(data35)
    sent = bits_of(blk[first:35]).astype(np.uint8)
    tx = np.arange(first, 35 + npar)
    bad = rng.choice(tx, size=min(n_err, tx.size), replace=False) if n_err else np.zeros(0, np.int64)
    rx = blk.copy()
    for p in bad:
        rx[p] ^= rng.integers(1, 64)
    bits = rng.integers(0, 2, 360).astype(np.uint8)
    llr = (rng.integers(strong - 40, strong + 40, 360) * rng.choice([-1, 1], 360)).astype(np.int16)
    pl, pa = bits_of(rx[first:35]), bits_of(rx[35:35 + npar])
    bits[XCCH_PAYLOAD_POS[kind]] = pl
    bits[XCCH_PARITY_POS[kind]] = pa
    sym_pos = {int(p): (XCCH_PAYLOAD_POS[kind][6 * (p - first):6 * (p - first) + 6] if p < 35
                        else XCCH_PARITY_POS[kind][6 * (p - 35):6 * (p - 35) + 6]) for p in tx}
    weak = [int(p) for p in (rng.choice(bad, size=min(weak_errors, len(bad)), replace=False) if weak_errors and len(bad) else [])]
    good = [int(p) for p in tx if p not in set(int(q) for q in bad)]
    weak += [int(p) for p in (rng.choice(good, size=min(n_weak_good, len(good)), replace=False) if n_weak_good else [])]
    for p in weak:
        b = sym_pos[p][int(rng.integers(0, 6))]
        llr[b] = int(rng.integers(0, 60)) * int(rng.choice([-1, 1]))
    return bits, llr, sent


def make_ess_case(rng, n_err, weak_errors, n_weak_good=0, strong=200):
    """-> (payload bits u8 [96], payload llr i16 [96], parity bits u8 [168], parity llr i16 [168], sent payload bits): a valid
    RS(44,16) ESS section with n_err corrupted symbols, weak_errors of them with a low |LLR|"""
    data35 = np.zeros(35, np.uint8)
    data35[19:35] = rng.integers(0, 64, 16)
    blk = encode_block(data35)
    sent = bits_of(blk[19:35]).astype(np.uint8)
    tx = np.arange(19, 63)
    bad = rng.choice(tx, size=min(n_err, 44), replace=False) if n_err else np.zeros(0, np.int64)
    rx = blk.copy()
    for p in bad:
        rx[p] ^= rng.integers(1, 64)
    pl, pa = bits_of(rx[19:35]).astype(np.uint8), bits_of(rx[35:63]).astype(np.uint8)
    llr = (rng.integers(strong - 40, strong + 40, 264) * rng.choice([-1, 1], 264)).astype(np.int16)
    weak = [int(p) for p in (rng.choice(bad, size=min(weak_errors, len(bad)), replace=False) if weak_errors and len(bad) else [])]
    good = [int(p) for p in tx if p not in set(int(q) for q in bad)]
    weak += [int(p) for p in (rng.choice(good, size=min(n_weak_good, len(good)), replace=False) if n_weak_good else [])]
    for p in weak:
        llr[6 * (p - 19) + int(rng.integers(0, 6))] = int(rng.integers(0, 60)) * int(rng.choice([-1, 1]))
    return pl, llr[:96].copy(), pa, llr[96:].copy(), sent
