"""CPU: Gardner timing-recovery restatement vs golden vectors from the reference's op25_gardner_cc, and (where the
compiled reference is present) vs the reference itself on fresh inputs and arbitrary block cuts."""
import numpy as np
import pytest

import orc
from conftest import golden


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["p25_cqpsk_48k", "p25_cqpsk_24k", "p25p2_48k"])
def test_gardner_golden(built, name):
    g = golden("ted_gardner.npz")
    sps, rate = [int(x) for x in g[name + "_cfg"]]
    iq = g[name + "_iq"]
    t = orc.OracleTed(sps, rate)
    outs, pos = [], 0
    for b in g[name + "_blocks"]:
        m = min(int(b), iq.shape[0] - pos)
        if m <= 0:
            break
        outs.append(t.block(iq[pos:pos + m]))
        pos += m
    got = np.concatenate(outs, axis=0)
    assert got.shape == g[name + "_sym"].shape
    assert np.array_equal(bits(got), bits(g[name + "_sym"]))
    # timing loop converged: symbol period within the +-0.2 % clamp, one symbol per ~sps samples
    assert abs(got.shape[0] - iq.shape[0] / sps) < 0.01 * iq.shape[0] / sps + 3


def test_stream_cut_invariance(built):
    """Results do not depend on how the stream is cut into calls (blocks of >= 4 samples)."""
    iq = orc.synth_qpsk_f32(5, 1, 600, 10)[0]
    a = orc.OracleTed(10, 4800).block(iq)
    t = orc.OracleTed(10, 4800)
    parts = [t.block(iq[i:i + 777]) for i in range(0, iq.shape[0], 777)]
    assert np.array_equal(bits(a), bits(np.concatenate(parts, axis=0)))


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_gardner_vs_compiled_reference(built):
    for sps, rate, gain in ((10, 4800, 0.0), (4, 6000, 0.03), (20, 2400, 0.0)):
        iq = orc.synth_qpsk_f32(100 + sps, 1, 900, sps, noise=0.2)[0]
        blocks = [5, 1000, 33, 4096, 10 ** 9]
        want, st = orc.ref_ted_blocks(iq, sps, rate, gain, blocks)
        t = orc.OracleTed(sps, rate, gain)
        pos = 0
        for b, w in zip(blocks, want):
            m = min(b, iq.shape[0] - pos)
            got = t.block(iq[pos:pos + m])
            assert np.array_equal(bits(got), bits(w))
            pos += m
