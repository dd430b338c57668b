"""The configs[2] chain on the CPU: one channel of cu8 I/Q through the oracles (front end -> receive loop -> NID BCH ->
TSDU trellis + CRC / LDU voice frames -> IMBE frame decode -> synthesis).  TEST INFRASTRUCTURE: the checker of the
end-to-end tests and of bench.py's parity gate, and the thing bench.py times as `cpu_baseline`."""
import ctypes as C
import time

import numpy as np

import ddn
import fecgen
import mbe
import orc
import p25gen

_NID_KEEP = [k for k in range(33) if k != 11]
_BP = None


def _layout():
    global _BP
    if _BP is None:
        first9 = np.zeros(9, np.int32)
        st9 = np.zeros(9, np.int32)
        ddn.lib().ddn_p25p1_layout_ldu_imbe(first9.ctypes.data, st9.ctypes.data)
        _BP = (first9, st9, np.array(p25gen.block_positions()) - 24)
    return _BP


def oracle_nid(bits, rel, obs, par, prel):
    from test_oracle_block import oracle_nid as f
    return f(bits, rel, obs, par, prel)


def run_channel(iq_c, lock, Fv, seed, state=None, timers=None, use_ref_front_end=False):
    """One call's worth of one channel.  `state` carries {fe, rx, voc} across calls (streaming).  With `timers` (dict) the
    seconds spent inside the C calls of each stage are accumulated there (Python glue excluded).
    -> dict(disc, sym, rec4, fl, acc, nids, tsbk, tsbk_crc, n_ldu, imbe_d, imbe_res, skip, pcm)"""
    first9, st9, bp = _layout()
    if state is None:
        state = {}
    T = timers if timers is not None else {}

    def timed(key, fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        T[key] = T.get(key, 0.0) + (time.perf_counter() - t0)
        return r

    if use_ref_front_end and orc.have_ref():
        disc = timed("front_end", orc.ref_front_end_cu8, iq_c, 8192)[0]       # stateless helper: whole-call only
    else:
        fe = state.setdefault("fe", orc.OracleFrontEnd())
        disc = timed("front_end", fe.run_cu8, iq_c, 8192)
    rx = state.setdefault("rx", orc.OracleP25Rx(lock_symbols=int(lock), use_filter=1))
    sym, rec4, fl = timed("rx", rx.run, disc)
    cnt = len(sym)
    acc = np.flatnonzero(fl[:cnt] & 2)
    # NIDs of every sync whose 33 NID dibits lie inside the call (what the framer gathers)
    ok = [a for a in acc if a + 34 <= cnt]
    nids = [None] * len(acc)
    if ok:
        nd = np.stack([rec4[a + 1:a + 34][_NID_KEEP] for a in ok])                               # [k, 32, 4]
        b = np.stack([(nd[:, :, 0] >> 1) & 1, nd[:, :, 0] & 1], axis=2).reshape(len(ok), 64).astype(np.uint8)
        r = np.minimum(np.abs(np.stack([nd[:, :, 2], nd[:, :, 3]], axis=2)), 255).reshape(len(ok), 64).astype(np.uint8)
        dec = timed("nid", oracle_nid, np.ascontiguousarray(b[:, :63]), np.ascontiguousarray(r[:, :63]),
                    np.zeros(len(ok), np.int32), np.ascontiguousarray(b[:, 63]), np.ascontiguousarray(r[:, 63]))
        for i, a in enumerate(ok):
            nids[list(acc).index(a)] = dec[i]
    # TSDU path: first trellis block of every frame that is complete
    tsbk, tsbk_ok = {}, {}
    full = [i for i, a in enumerate(acc) if a + 1 + bp[-1] < cnt]
    if full:
        llr = np.stack([np.stack([rec4[acc[i] + 1 + bp, 2], rec4[acc[i] + 1 + bp, 3]], axis=1).reshape(196) for i in full]).astype(np.int16)
        blocks, _ = timed("trellis", fecgen.oracle_p25_half_rate, np.ascontiguousarray(llr))
        for j, i in enumerate(full):
            tsbk[i] = blocks[j]
            tsbk_ok[i] = p25gen.crc16_ccitt(blocks[j][:10]) == ((int(blocks[j][10]) << 8) | int(blocks[j][11]))
    # voice path
    frames, skip = [], []
    n_ldu = 0
    for i, a in enumerate(acc):
        n = nids[i]
        if n is None or n[0] != 1 or n[2] not in (5, 10) or n_ldu >= Fv:
            continue
        n_ldu += 1
        for v in range(9):
            s0 = a - 23 + int(first9[v])
            d = rec4[s0:min(s0 + 76, cnt)]
            fr, _, flag, _, _ = timed("imbe_deint", orc.oracle_imbe_deinterleave, d[:, 0].astype(np.uint8), d[:, 2].astype(np.int16),
                                      d[:, 3].astype(np.int16), int(st9[v]))
            frames.append(fr)
            skip.append(flag != 0)
    V = Fv * 9
    imbe_d = np.zeros((V, 88), np.uint8)
    imbe_res = np.zeros((V, 5), np.int32)
    skipv = np.ones(V, bool)
    if frames:
        bits, res, rc = timed("imbe_fec", mbe.oracle_frame_decode, ddn.MBE_IMBE, np.stack(frames))
        assert np.all(rc == 0)
        imbe_d[:len(frames)] = bits
        imbe_res[:len(frames)] = res
        skipv[:len(frames)] = skip
    voc = state.setdefault("voc", mbe.OracleVocoder(ddn.MBE_IMBE, 1, tail_rule=1))
    pcm = np.zeros((V, 160), np.float32)
    live = np.flatnonzero(~skipv)
    if live.size:
        lb = np.ascontiguousarray(imbe_d[live])
        lr = np.ascontiguousarray(imbe_res[live])
        lp = np.zeros((live.size, 160), np.float32)
        rc = timed("mbe_synth", mbe._o().om_process_batch, ddn.MBE_IMBE, C.addressof(voc.tab), lb.ctypes.data, lr.ctypes.data, 1,
                   int(seed), 1, int(live.size), lp.ctypes.data, None, C.addressof(voc.cur), C.addressof(voc.prev),
                   C.addressof(voc.enh))
        assert rc == 0
        pcm[live] = lp
    return dict(disc=disc, sym=sym, rec4=rec4, fl=fl, acc=acc, nids=nids, tsbk=tsbk, tsbk_crc=tsbk_ok, n_ldu=n_ldu,
                imbe_d=imbe_d, imbe_res=imbe_res, skip=skipv, pcm=pcm)
