"""CPU: the rational L/M polyphase resampler restatement (oracle/ddn_oracle_resamp.c) pinned bit for bit against the
reference's compiled dsd_resampler_design / dsd_resampler_process_block (src/dsp/resampler.cpp, oracle/_ref)."""
import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
RATIOS = [(1, 1), (2, 1), (1, 2), (5, 4), (4, 5), (3, 7), (10, 3), (160, 147), (147, 160), (25, 24)]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@needs_ref
@pytest.mark.parametrize("L,M", RATIOS)
def test_resampler_taps_and_stream(L, M):
    ref, o = orc.RefResampler(L, M), orc.OracleResampler(L, M)
    assert np.array_equal(bits(ref.taps()), bits(o.taps()))
    rng = np.random.default_rng(FZ + L * 1000 + M)
    x = (rng.normal(0, 9000, 5000) + 12000 * np.sin(np.arange(5000) * 0.07)).astype(np.float32)
    cuts = [0, 1, 2, 17, 18, 700, 701, 3333, 5000]   # ragged blocks incl. shorter than the 16-tap window
    for a, b in zip(cuts[:-1], cuts[1:]):
        want, got = ref.run(x[a:b]), o.run(x[a:b])
        assert len(want) == len(got) and np.array_equal(bits(want), bits(got)), (L, M, a, b)
    assert len(ref.run(x[:0])) == 0 and len(o.run(x[:0])) == 0
