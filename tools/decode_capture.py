#!/usr/bin/env python3
"""Decode a dsd-neo I/Q capture (P25 Phase 1, C4FM) with libdsdneo_hip.so and print a per-frame summary - the same facts the
reference's --iq-replay log carries (NAC, DUID, TSBK opcodes with CRC status, LDU1 link control, LDU2 encryption sync,
voice-frame counts).  Everything between the capture file and the printed lines runs through the C-ABI: capture reader,
front end, receive loop, framer gathers, BCH / trellis / CRC / Hamming / Reed-Solomon kernels.

usage: python tools/decode_capture.py CAPTURE.iq[.json] [--lock SYMBOLS] [--max-frames N]
       --lock: in-frame symbols after a sync (default 840 = LDU-sized; 156 / 336 suit one- / three-block TSDU control
               channels - dibits read outside the in-frame span carry no soft decisions)"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
DUID = {0: "HDU", 3: "TDU", 5: "LDU1", 7: "TSBK", 10: "LDU2", 12: "PDU", 15: "TDULC"}


def decode(path, lock=840, max_frames=64, out=print):
    import numpy as np
    import torch
    import ddn
    l = ddn.lib()
    dev = lambda *sh, dt=torch.uint8: torch.zeros(sh, dtype=dt, device="cuda")
    # ---- capture -> channel-major batch of one
    paths = (C.c_char_p * 1)(path.encode())
    buf, n = C.c_void_p(), C.c_size_t()
    info = (C.c_uint8 * 8192)()
    rc = l.ddn_iq_load_batch(paths, 1, C.byref(buf), C.byref(n), info)
    if rc != 0:
        raise SystemExit("cannot open capture (%d): %s" % (rc, l.ddn_last_error().decode()))
    fmt, rate, base_dec, _, demod_rate = np.frombuffer(bytes(info[:24]), np.uint32)[1:6]
    bps = {1: 2, 2: 8}.get(int(fmt))
    if bps is None:
        raise SystemExit("sample format %d is not supported by the front end (cu8 / cf32 only)" % fmt)
    n = n.value
    host = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), (n * bps,)).copy()
    l.ddn_iq_free(buf)
    out("capture: %d complex samples, %s @ %d Hz, base_decimation %d -> demod %d Hz"
        % (n, "cu8" if fmt == 1 else "cf32", rate, base_dec, demod_rate))
    passes = int(base_dec).bit_length() - 1
    n -= n % (1 << passes)
    fe = ddn.Batch(1, sample_rate_hz=int(demod_rate), input_format=ddn.IN_CU8 if fmt == 1 else ddn.IN_CF32,
                   block_len=8192)
    if passes:
        fe.set_decimation(passes)
    d_iq = torch.from_numpy(host[:n * bps]).cuda()
    nd = n >> passes
    d_disc = dev(1, nd, dt=torch.float32)
    fe.run_device(d_iq.data_ptr(), n, d_disc.data_ptr())
    # ---- discriminator -> records -> frame slots
    rx = ddn.P25Rx(1, out_rate=int(demod_rate), lock_symbols=lock, use_matched_filter=1 if demod_rate == 48000 else 0)
    ms = l.ddn_p25_rx_max_symbols(rx.h, nd)
    rec, fl, cnt = dev(1, ms, 10), dev(1, ms), dev(1, dt=torch.int32)
    assert l.ddn_p25_rx_run(rx.h, d_disc.data_ptr(), nd, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    F = max_frames
    fr = C.c_void_p()
    assert l.ddn_p25p1_framer_create(1, F, C.byref(fr)) == 0
    assert l.ddn_p25p1_framer_index(fr, fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    args = (rec.data_ptr(), cnt.data_ptr(), ms)
    bits, rel, par, prel, nid = dev(F, 63), dev(F, 63), dev(F), dev(F), dev(F, 4, dt=torch.int32)
    obs = dev(F, dt=torch.int32)
    assert l.ddn_p25p1_framer_gather_nid(fr, *args, bits.data_ptr(), rel.data_ptr(), par.data_ptr(), prel.data_ptr(), None,
                                         None) == 0
    assert l.ddn_p25p1_nid_decode_batch(bits.data_ptr(), rel.data_ptr(), obs.data_ptr(), par.data_ptr(), prel.data_ptr(), 64,
                                        F, nid.data_ptr(), None) == 0
    # trunking blocks
    tsbk, tsbk_ok, tsbk_v = [], [], []
    for b in range(3):
        llr, v = dev(F, 196, dt=torch.int16), dev(F)
        o12, met, ok = dev(F, 12), dev(F, dt=torch.int32), dev(F)
        assert l.ddn_p25p1_framer_gather_trellis_block(fr, b, *args, llr.data_ptr(), None, v.data_ptr(), None) == 0
        assert l.ddn_fec_p25_12_soft_batch(llr.data_ptr(), F, o12.data_ptr(), met.data_ptr(), None) == 0
        assert l.ddn_fec_p25_crc16_batch(o12.data_ptr(), 12, F, ok.data_ptr(), None) == 0
        tsbk.append(o12)
        tsbk_ok.append(ok)
        tsbk_v.append(v)
    # LDU words -> Hamming -> RS
    ldu = {}
    for which, code, nd_ in ((1, 0, 12), (2, 1, 16)):
        w, v, errs = dev(F, 240), dev(F), dev(F * 24)
        d6, p6, st = dev(F, nd_, 6), dev(F, 24 - nd_, 6), dev(F)
        assert l.ddn_p25p1_framer_gather_ldu_words(fr, which, *args, w.data_ptr(), None, v.data_ptr(), None) == 0
        assert l.ddn_fec_hamming_10_6_3_batch(w.data_ptr(), F * 24, errs.data_ptr(), None) == 0
        assert l.ddn_p25p1_framer_pack_ldu_rs(fr, which, w.data_ptr(), d6.data_ptr(), p6.data_ptr(), None) == 0
        assert l.ddn_fec_p25_rs_batch(code, d6.data_ptr(), p6.data_ptr(), F, st.data_ptr(), None) == 0
        ldu[which] = (d6, st, v)
    # voice frames
    first, sc = dev(F * 9, dt=torch.int64), dev(F * 9, dt=torch.int32)
    ifr, isf, ifl, isc = dev(F * 9, 8, 23), dev(F * 9, 8, 23, 2), dev(F * 9), dev(F * 9, dt=torch.int32)
    assert l.ddn_p25p1_framer_imbe_index(fr, ms, first.data_ptr(), sc.data_ptr(), None) == 0
    assert l.ddn_p25p1_imbe_deinterleave_batch(rec.data_ptr(), ms, first.data_ptr(), sc.data_ptr(), F * 9, ifr.data_ptr(),
                                               isf.data_ptr(), ifl.data_ptr(), isc.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ns = np.zeros(1, np.int32)
    pos = np.zeros(F, np.int32)
    assert l.ddn_p25p1_framer_get_syncs(fr, ns.ctypes.data, pos.ctypes.data) == 0
    l.ddn_p25p1_framer_destroy(fr)
    nidh = nid.cpu().numpy()
    val = lambda b: int("".join(str(int(x)) for x in b), 2)
    lines = []
    for k in range(int(ns[0])):
        st_, nac, duid, errs = (int(x) for x in nidh[k])
        head = "sync @%6d  " % pos[k]
        if st_ != 1:
            lines.append(head + "NID undecodable")
            continue
        text = head + "NAC %03X  %-5s" % (nac, DUID.get(duid, "DUID%X" % duid)) + ("  (NID %d bit fixes)" % errs if errs else "")
        if duid in (7, 12):
            for b in range(3):
                if not int(tsbk_v[b][k]):
                    break
                by = tsbk[b][k].cpu().numpy()
                text += "  | blk%d op %02X mfid %02X %s" % (b, by[0] & 0x3F, by[1], "crc ok" if int(tsbk_ok[b][k]) else "CRC ERR")
                if (by[0] & 0x3F) == 0x3B and int(tsbk_ok[b][k]):
                    text += " NET_STS WACN %05X SYS %03X" % ((int(by[3]) << 12) | (int(by[4]) << 4) | (int(by[5]) >> 4),
                                                            ((int(by[5]) & 0xF) << 8) | int(by[6]))
                if by[0] & 0x80:
                    break                                  # last block flag
        elif duid == 5 and int(ldu[1][2][k]):
            d6 = ldu[1][0][k].cpu().numpy()
            lc = d6[::-1].reshape(72)
            ok = int(ldu[1][1][k]) == 0
            text += "  | LC %s: LCO %02X MFID %02X" % ("ok" if ok else "RS ERR", val(lc[2:8]), val(lc[8:16]))
            if ok and val(lc[2:8]) == 0 and val(lc[8:16]) in (0, 1):
                text += "  Group Voice Channel User TG %d SRC %d" % (val(lc[32:48]), val(lc[48:72]))
            text += "  | 9 IMBE frames (%d flagged)" % int((ifl[9 * k:9 * k + 9] != 0).sum())
        elif duid == 10 and int(ldu[2][2][k]):
            hx = ldu[2][0][k].cpu().numpy()
            ok = int(ldu[2][1][k]) == 0
            text += "  | ESS %s: ALGID %02X KID %04X" % ("ok" if ok else "RS ERR", val(list(hx[3]) + list(hx[2][:2])),
                                                          val(list(hx[2][2:]) + list(hx[1]) + list(hx[0])))
            text += "  | 9 IMBE frames (%d flagged)" % int((ifl[9 * k:9 * k + 9] != 0).sum())
        lines.append(text)
    for t in lines:
        out(t)
    return lines


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("capture")
    ap.add_argument("--lock", type=int, default=840)
    ap.add_argument("--max-frames", type=int, default=64)
    a = ap.parse_args()
    decode(a.capture, a.lock, a.max_frames)


if __name__ == "__main__":
    main()
