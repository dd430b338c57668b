"""Differential fuzz of the DMR / NXDN48 receive loop: random traffic built from the committed captures (level, polarity, noise,
gaps long enough for carrier loss, silence), random batch shape, protocol, modulation rules, polarity flag, matched filter,
handler lengths and call splits - GPU vs the CPU restatement, bit for bit.  DDN_FUZZ_BASE=k shifts the seeds."""
import os

import numpy as np
import pytest

import ddn
import rx4
from test_rx4_gpu import check_channel

pytestmark = pytest.mark.gpu
BASE = int(os.environ.get("DDN_FUZZ_BASE", "0"))
_DISC = {}


def disc(name, lpf):
    if name not in _DISC:
        _DISC[name] = rx4.capture_disc(name, lpf)
    return _DISC[name]


@pytest.mark.parametrize("case", range(16))
def test_fuzz_rx4(built, case):
    rng = np.random.default_rng(1000 * BASE + case + 77)
    nxdn = bool(rng.integers(0, 2))
    proto = rx4.PROTO_NXDN48 if nxdn else rx4.PROTO_DMR
    src = disc("iq_nxdn48.npz", 1)[60000:] if nxdn else disc(["iq_dmr_t3_ras_cc.npz", "iq_dmr_voice.npz"][int(rng.integers(0, 2))], 2)
    rf_mod = int(rng.choice([0, 2]))
    inv = int(rng.integers(0, 2)) if not nxdn else 0
    use_filter = int(rng.integers(0, 2))
    B = int(rng.choice([1, 3, 9, 17, 40]))
    n = int(rng.integers(9000, 60000))
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        pos = 0
        while pos < n:
            kind = rng.random()
            ln = int(rng.integers(500, 30000))
            ln = min(ln, n - pos)
            if kind < 0.6:
                o = int(rng.integers(0, len(src) - ln))
                seg = src[o:o + ln] * np.float32(rng.choice([1.0, -1.0, 0.4, 1.7]))
                seg = seg + rng.standard_normal(ln).astype(np.float32) * np.float32(rng.choice([0, 150, 900]))
            elif kind < 0.8:
                seg = rng.standard_normal(ln).astype(np.float32) * np.float32(rng.choice([50, 2000, 12000]))
            else:
                seg = np.zeros(ln, np.float32)
            x[c, pos:pos + ln] = seg
            pos += ln
    lock = np.zeros((B, 4), np.int32)
    for c in range(B):
        lock[c] = [int(rng.choice([0, 54, 120, 182, 700, 3000])), int(rng.choice([0, 54, 342, 1782])), 0, 0]
    gpu = ddn.Fsk4Rx(B, ddn.FSK4_NXDN48 if nxdn else ddn.FSK4_DMR, rf_mod=rf_mod, inverted=inv, use_matched_filter=use_filter)
    assert ddn.lib().ddn_fsk4_rx_set_lock_symbols(gpu.h, lock.ctypes.data) == 0
    # every kernel shape (the default for a batch this small is one channel per wavefront); drawn from its own stream so the traffic
    # of a case stays what it was
    cpw = int(np.random.default_rng(5000 * BASE + case).choice([1, 2, 4, 8, 16, 32]))
    assert ddn.lib().ddn_fsk4_rx_set_channels_per_wave(gpu.h, cpw) == 0
    cpu = [rx4.OracleFsk4Rx(rx4.profile(proto, rf_mod=rf_mod, use_filter=use_filter, inverted=inv, lock=[int(v) for v in lock[c]]))
           for c in range(B)]
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(1, n, int(rng.integers(0, 4)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        got = gpu.run_host(x[:, a:b])
        for c in range(B):
            check_channel(got, c, cpu[c].run(x[c, a:b], max_sync=got["sync_pos"].shape[1]))
