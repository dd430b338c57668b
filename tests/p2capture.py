"""Test scaffolding: the reference's P25 Phase 2 capture (tests/golden/iq_p25p2_cc.npz == tests/fixtures/iq/p25p2_cc.iq, 48 kHz cu8, the
signal DECODE_IQ_P25P2_CC feeds to `-f2`, tests/CMakeLists.txt:8923) brought down to timeslots of 360 bits + soft metrics with a plain
numpy H-DQPSK demodulator (6000 symbols/s = 8 samples per symbol: 5-tap average, differential phase over one symbol, the quadrant is the
dibit, the distance from the quadrant boundary the metric).  Not the reference's CQPSK receiver and not the product's front end (which is
the C4FM / FSK4 path): it only has to deliver the capture's dibits, and it does so without an error - every S-ISCH word, I-ISCH word and
DUID word on it is an exact code word, and the ten scrambled SACCH bursts decode with no symbol to correct.

What the capture holds (measured here, see tests/test_oracle_p25p2_capture.py): a TDMA channel of WACN BEE00, system 164, NAC 161 - the
44-bit scrambler seed is not in the 0.25 s of signal as a message; it was recovered once from the SACCH bursts themselves (received =
code word + LFSR sequence is linear over GF(2) in the 180 payload bits and the 44 seed bits: 312 equations, 224 unknowns, consistent for
exactly one seed across the bursts) and is a known answer of the fixture since."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
S_ISCH = np.array([1, 1, 1, 3, 1, 1, 3, 1, 1, 1, 1, 3, 3, 3, 1, 3, 3, 3, 3, 3], np.uint8)      # 0x575D57F7FF as dibits
WACN, SYSID, NAC = 0xBEE00, 0x164, 0x161
DUID_OFFSETS = [0, 1, 74, 75, 244, 245, 318, 319]                                                  # p25p2_frame.c:1464


def dibits():
    iq = np.load(os.path.join(HERE, "golden", "iq_p25p2_cc.npz"))["iq"].astype(np.float64) - 127.5
    x = np.convolve(iq[:, 0] + 1j * iq[:, 1], np.ones(5) / 5, mode="same")
    s = np.angle(x[8:] * np.conj(x[:-8]))[0::8]
    dib = np.where(s >= 0, np.where(s < np.pi / 2, 0, 1), np.where(s > -np.pi / 2, 2, 3)).astype(np.uint8)
    rel = np.minimum(np.abs(np.abs(s) - np.pi / 2) * 400, 255).astype(np.int16)
    return dib, rel


def timeslots():
    """-> (bits u8 [n][360], llr i16 [n][360] (magnitudes: the sign convention of p2llr plays no part in what is tested), slot of the
    superframe i32 [n], sync hits): timeslot = the 160 dibits of a burst + the 20 dibits of the ISCH that follows it
    (p2_dibit_buffer() starts right behind the S-ISCH it synchronised on, p25p2_frame.c:354-372)"""
    dib, rel = dibits()
    hits = [i for i in range(len(dib) - 20) if (dib[i:i + 20] != S_ISCH).sum() <= 1]
    t0 = (hits[0] + 20) % 180
    ts = list(range(t0, len(dib) - 180, 180))
    bits, llr = np.zeros((len(ts), 360), np.uint8), np.zeros((len(ts), 360), np.int16)
    for k, t in enumerate(ts):
        d = dib[t:t + 180]
        bits[k, 0::2], bits[k, 1::2] = d >> 1, d & 1
        llr[k] = np.repeat(np.maximum(rel[t:t + 180], 1), 2)
    # the timeslot whose I-ISCH reads channel 1, location 0 is slot 0 of the superframe (p25p2_process_isch(), p25p2_frame.c:728-736:
    # p2_scramble_offset = 12 - framing_counter there) - on this capture the 11th timeslot
    sf = (np.arange(len(ts)) + 2) % 12
    return bits, llr, sf.astype(np.int32), hits
