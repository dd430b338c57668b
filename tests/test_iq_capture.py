"""dsd-neo I/Q capture reader (ddn_iq_capture_*) vs the reference's own reader
(dsd_iq_replay_read_metadata / _open / _read of src/io/iq/iq_replay.c, compiled in place into oracle/_ref): every field,
the replayable-byte rule, the data bytes delivered, and the error code for each kind of broken sidecar.  Host-only code,
runs without a GPU; the sweep over the reference's 16 fixture captures runs where /root/reference exists."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

import ddn
import orc
from conftest import golden

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
FIXTURES = sorted(glob.glob("/root/reference/tests/fixtures/iq/*.iq.json"))


class Info(C.Structure):
    _fields_ = [("metadata_version", C.c_uint32), ("sample_format", C.c_int), ("sample_rate_hz", C.c_uint32),
                ("base_decimation", C.c_uint32), ("post_downsample", C.c_uint32), ("demod_rate_hz", C.c_uint32),
                ("capture_retune_count", C.c_uint32), ("event_count", C.c_uint32), ("center_frequency_hz", C.c_uint64),
                ("capture_center_frequency_hz", C.c_uint64), ("data_bytes", C.c_uint64), ("capture_drops", C.c_uint64),
                ("capture_drop_blocks", C.c_uint64), ("input_ring_drops", C.c_uint64), ("actual_file_bytes", C.c_uint64),
                ("effective_bytes", C.c_uint64), ("ppm", C.c_int), ("tuner_gain_tenth_db", C.c_int),
                ("rtl_dsp_bw_khz", C.c_int), ("offset_tuning_enabled", C.c_int), ("fs4_shift_enabled", C.c_int),
                ("combine_rotate_enabled", C.c_int), ("muted_bytes_excluded", C.c_int), ("contains_retunes", C.c_int),
                ("size_limit_reached", C.c_int), ("size_mismatch", C.c_int), ("capture_stage", C.c_char * 64),
                ("data_path", C.c_char * 2048), ("metadata_path", C.c_char * 2048)]


class Event(C.Structure):
    _fields_ = [("kind", C.c_int), ("byte_offset", C.c_uint64), ("duration_bytes", C.c_uint64),
                ("center_frequency_hz", C.c_uint64), ("capture_center_frequency_hz", C.c_uint64),
                ("sample_rate_hz", C.c_uint32), ("reason", C.c_char * 64)]


def ref_meta(path, for_replay=0):
    r = orc.ref()
    r.refh_iq_meta.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    out = np.zeros(23, np.int64)
    stage, dp, err = C.create_string_buffer(64), C.create_string_buffer(2048), C.create_string_buffer(256)
    rc = r.refh_iq_meta(path.encode(), for_replay, out.ctypes.data, stage, dp, err)
    return rc, out, stage.value, dp.value, err.value


def mine(path, for_replay=0):
    l = ddn.lib()
    if not for_replay:
        info = Info()
        return l.ddn_iq_capture_read_info(path.encode(), C.byref(info)), info
    h = C.c_void_p()
    rc = l.ddn_iq_capture_open(path.encode(), C.byref(h))
    info = Info()
    if rc == 0:
        C.memmove(C.byref(info), l.ddn_iq_capture_get_info(h), C.sizeof(Info))
        l.ddn_iq_capture_close(h)
    return rc, info


def same_fields(info, out, stage, dp):
    got = [info.metadata_version, info.sample_format, info.sample_rate_hz, info.base_decimation, info.post_downsample,
           info.demod_rate_hz, info.capture_retune_count, info.event_count, info.center_frequency_hz,
           info.capture_center_frequency_hz, info.data_bytes, info.capture_drops, info.capture_drop_blocks,
           info.input_ring_drops, info.ppm, info.tuner_gain_tenth_db, info.rtl_dsp_bw_khz, info.offset_tuning_enabled,
           info.fs4_shift_enabled, 0 if info.combine_rotate_enabled else 1, info.muted_bytes_excluded,
           info.contains_retunes, info.size_limit_reached]
    assert got == list(out), (got, list(out))
    assert info.capture_stage == stage and info.data_path == dp


BASE = {"format": "dsd-neo-iq", "version": 1, "sample_format": "cu8", "iq_order": "IQ", "endianness": "none",
        "capture_stage": "post_mute_pre_widen", "sample_rate_hz": 96000, "center_frequency_hz": 851375000,
        "capture_center_frequency_hz": 851375000, "ppm": -3, "tuner_gain_tenth_db": 270, "rtl_dsp_bw_khz": 48,
        "base_decimation": 2, "post_downsample": 1, "demod_rate_hz": 48000, "offset_tuning_enabled": False,
        "fs4_shift_enabled": True, "combine_rotate_enabled": False, "muted_bytes_excluded": True,
        "contains_retunes": False, "capture_retune_count": 0, "source_backend": "rtl", "source_args": "dev=0",
        "capture_started_utc": "2026-07-30T00:00:00Z", "data_file": "cap.iq", "data_bytes": 2000, "capture_drops": 1,
        "capture_drop_blocks": 2, "input_ring_drops": 3, "notes": 'a "quoted" note', "extra_unknown": 7, "another": "x", "third": None}


def write_capture(tmp, meta, data_len=2001, name="cap.iq"):
    rng = np.random.default_rng(0)
    rng.integers(0, 256, data_len, dtype=np.uint8).tofile(os.path.join(tmp, name))
    p = os.path.join(tmp, name + ".json")
    with open(p, "w") as f:
        f.write(meta if isinstance(meta, str) else json.dumps(meta, indent=1))
    return p


@needs_ref
@pytest.mark.skipif(not FIXTURES, reason="/root/reference not present")
@pytest.mark.parametrize("meta_path", FIXTURES, ids=[os.path.basename(p)[:-8] for p in FIXTURES])
def test_reference_fixture_captures(built, meta_path):
    rc_r, out, stage, dp, _ = ref_meta(meta_path)
    rc_m, info = mine(meta_path)
    assert rc_r == 0 and rc_m == 0
    same_fields(info, out, stage, dp)
    data_path = meta_path[:-5]
    rc2, info2 = mine(data_path)                     # the data path is accepted as well (sidecar = path + ".json")
    assert rc2 == 0 and info2.data_bytes == info.data_bytes
    size = os.path.getsize(data_path)
    eff = C.c_uint64()
    mm = C.c_int()
    orc.ref().dsd_iq_replay_compute_effective_bytes.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    assert orc.ref().dsd_iq_replay_compute_effective_bytes(info.data_bytes, size, info.sample_format, C.byref(eff),
                                                           C.byref(mm)) == 0
    assert (info.effective_bytes, info.size_mismatch, info.actual_file_bytes) == (eff.value, mm.value, size)
    # bytes delivered by the two readers, odd request sizes included
    l = ddn.lib()
    orc.ref().refh_iq_read_all.restype = C.c_long
    orc.ref().refh_iq_read_all.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.c_int]
    want = np.zeros(size + 16, np.uint8)
    n_ref = orc.ref().refh_iq_read_all(meta_path.encode(), want.ctypes.data, want.size, 4099)
    h = C.c_void_p()
    assert l.ddn_iq_capture_open(meta_path.encode(), C.byref(h)) == 0
    got = np.zeros(size + 16, np.uint8)
    total, k = 0, C.c_size_t()
    while True:
        assert l.ddn_iq_capture_read(h, got.ctypes.data + total, 4099, C.byref(k)) == 0
        if k.value == 0:
            break
        total += k.value
    assert total == n_ref == info.effective_bytes and np.array_equal(got[:total], want[:total])
    assert l.ddn_iq_capture_rewind(h) == 0
    assert l.ddn_iq_capture_read(h, got.ctypes.data, 64, C.byref(k)) == 0 and np.array_equal(got[:64], want[:64])
    l.ddn_iq_capture_close(h)


@needs_ref
def test_synthetic_sidecars_match_reference(built, tmp_path):
    tmp = str(tmp_path)
    p = write_capture(tmp, BASE)
    for replay in (0, 1):
        rc_r, out, stage, dp, _ = ref_meta(p, replay)
        rc_m, info = mine(p, replay)
        assert rc_r == 0 and rc_m == 0
        same_fields(info, out, stage, dp)
    assert info.effective_bytes == 2000 and info.size_mismatch == 1       # 2001 bytes on disk, 2000 declared
    cases = {
        "no_format": {k: v for k, v in BASE.items() if k != "format"},
        "no_notes": {k: v for k, v in BASE.items() if k != "notes"},
        "bad_format": dict(BASE, format="other"),
        "bad_version": dict(BASE, version=3),
        "bad_order": dict(BASE, iq_order="QI"),
        "bad_endian": dict(BASE, endianness="little"),
        "cf32_none": dict(BASE, sample_format="cf32"),
        "cs16": dict(BASE, sample_format="cs16", endianness="little"),
        "fmt_unknown": dict(BASE, sample_format="cs8"),
        "dec_not_pow2": dict(BASE, base_decimation=3, demod_rate_hz=32000),
        "dec_too_big": dict(BASE, base_decimation=2048, sample_rate_hz=2048 * 48000),
        "rate_chain": dict(BASE, demod_rate_hz=47999),
        "zero_rate": dict(BASE, sample_rate_hz=0),
        "zero_post": dict(BASE, post_downsample=0),
        "bad_stage": dict(BASE, capture_stage="somewhere"),
        "events_in_v1": dict(BASE, events=[]),
        "retunes_no_timeline": dict(BASE, contains_retunes=True, capture_retune_count=2),
        "v2_events": dict(BASE, version=2, contains_retunes=True, capture_retune_count=1, events=[
            {"kind": "MUTE", "byte_offset": 100, "reason": "squelch", "duration_bytes": 64},
            {"kind": "RETUNE", "byte_offset": 400, "reason": "trunk", "center_frequency_hz": 852000000,
             "capture_center_frequency_hz": 852000000, "sample_rate_hz": 96000},
            {"kind": "RESET", "byte_offset": 400, "reason": "trunk", "center_frequency_hz": 852000000,
             "capture_center_frequency_hz": 852000000, "sample_rate_hz": 96000}]),
        "v2_retune_without_reset": dict(BASE, version=2, contains_retunes=True, capture_retune_count=1, events=[
            {"kind": "RETUNE", "byte_offset": 400, "reason": "trunk", "center_frequency_hz": 852000000,
             "capture_center_frequency_hz": 852000000, "sample_rate_hz": 96000}]),
        "v2_retune_count_mismatch": dict(BASE, version=2, contains_retunes=True, capture_retune_count=2, events=[
            {"kind": "RETUNE", "byte_offset": 400, "reason": "r", "center_frequency_hz": 1, "capture_center_frequency_hz": 1,
             "sample_rate_hz": 96000},
            {"kind": "RESET", "byte_offset": 400, "reason": "r", "center_frequency_hz": 1, "capture_center_frequency_hz": 1,
             "sample_rate_hz": 96000}]),
        "v2_event_rate_change": dict(BASE, version=2, events=[
            {"kind": "RESET", "byte_offset": 0, "reason": "r", "center_frequency_hz": 1, "capture_center_frequency_hz": 1,
             "sample_rate_hz": 48000}]),
        "v2_event_without_reason": dict(BASE, version=2, events=[{"kind": "MUTE", "byte_offset": 100, "duration_bytes": 2}]),
        "v2_misaligned_offset": dict(BASE, version=2, events=[{"kind": "MUTE", "byte_offset": 101, "reason": "r", "duration_bytes": 2}]),
        "v2_offset_past_end": dict(BASE, version=2, events=[{"kind": "MUTE", "byte_offset": 4000, "reason": "r", "duration_bytes": 2}]),
        "v2_zero_mute": dict(BASE, version=2, events=[{"kind": "MUTE", "byte_offset": 100, "reason": "r", "duration_bytes": 0}]),
        "v2_bad_kind": dict(BASE, version=2, events=[{"kind": "JUMP", "byte_offset": 0, "reason": "r"}]),
        "v2_mute_no_duration": dict(BASE, version=2, events=[{"kind": "MUTE", "byte_offset": 0, "reason": "x"}]),
        "v2_unordered": dict(BASE, version=2, events=[{"kind": "MUTE", "byte_offset": 200, "reason": "r", "duration_bytes": 2},
                                                      {"kind": "MUTE", "byte_offset": 100, "reason": "r", "duration_bytes": 2}]),
        "nested_unknown": dict(BASE, extra={"x": [1, 2]}),
        "not_json": "this is not json",
        "truncated": json.dumps(BASE)[:200],
        "float_rate": json.dumps(BASE).replace("96000", "96000.5"),
    }
    for name, meta in cases.items():
        d = os.path.join(tmp, name)
        os.mkdir(d)
        q = write_capture(d, meta)
        for replay in (0, 1):
            rc_r, out, stage, dp, err = ref_meta(q, replay)
            rc_m, info = mine(q, replay)
            assert rc_m == rc_r, (name, replay, rc_m, rc_r, err, ddn.lib().ddn_last_error())
            if rc_r == 0:
                same_fields(info, out, stage, dp)
    # events of the v2 case, field by field
    q = os.path.join(tmp, "v2_events", "cap.iq.json")
    l = ddn.lib()
    h = C.c_void_p()
    assert l.ddn_iq_capture_open(q.encode(), C.byref(h)) == 0
    n = C.c_uint32()
    ev = C.cast(l.ddn_iq_capture_get_events(h, C.byref(n)), C.POINTER(Event))
    assert n.value == 3 and ev[0].reason == b"squelch"
    orc.ref().refh_iq_event.argtypes = [C.c_char_p, C.c_uint, C.c_void_p]
    for i in range(3):
        o = np.zeros(6, np.int64)
        assert orc.ref().refh_iq_event(q.encode(), i, o.ctypes.data) == 0
        assert [ev[i].kind, ev[i].byte_offset, ev[i].duration_bytes, ev[i].center_frequency_hz,
                ev[i].capture_center_frequency_hz, ev[i].sample_rate_hz] == list(o)
    l.ddn_iq_capture_close(h)
    # missing sidecar / missing data file / empty data file
    assert mine(os.path.join(tmp, "nope.iq"))[0] == ref_meta(os.path.join(tmp, "nope.iq"))[0] != 0
    d = os.path.join(tmp, "nodata")
    os.mkdir(d)
    q = write_capture(d, BASE)
    os.remove(os.path.join(d, "cap.iq"))
    assert mine(q, 1)[0] == ref_meta(q, 1)[0] != 0
    open(os.path.join(d, "cap.iq"), "wb").close()
    assert mine(q, 1)[0] == ref_meta(q, 1)[0] != 0


GRAMMAR_CASES = [  # (edit of the BASE sidecar text, expected code of a metadata-only read); found by DDN_FUZZ_BASE sweeps
    (lambda t: t.replace('"data_bytes": 2000', '"data_bytes": 02000'), -2),         # no leading zeros in numbers
    (lambda t: t.replace('"data_bytes": 2000', '"data_bytes": 2000.0'), -2),        # integers only
    (lambda t: t.replace('"ppm": -3', '"ppm": -'), -2),
    (lambda t: t.replace('"ppm": -3', '"ppm": -03'), -2),
    (lambda t: t.replace("post_mute_pre_widen", "post_mute_p\\re_widen"), -2),      # escaped control byte: invalid, not "unsupported stage"
    (lambda t: t.replace("dev=0", "dev=\\u0007"), -2),
    (lambda t: t.replace('a \\"quoted\\" note', "line\\nbreak"), 0),                 # ... but "notes" may hold them
    (lambda t: t.replace('"notes": "a \\"quoted\\" note"', '"notes": null'), 0),
    (lambda t: t.replace('"notes": "a \\"quoted\\" note"', '"notes": "' + "n" * 256 + '"'), -2),   # 255 characters fit
    (lambda t: t.replace('"notes": "a \\"quoted\\" note"', '"notes": "' + "n" * 255 + '"'), 0),
    (lambda t: t.replace('"source_backend": "rtl"', '"source_backend": "' + "b" * 32 + '"'), -2),
    (lambda t: t.rstrip()[:-1].rstrip() + ",\n}", 0),                             # trailing comma before the closing brace
    (lambda t: t + " x", -2),                                                       # trailing content
    (lambda t: t + "\n\t \r", 0),
    (lambda t: t + "\f", -2),                                                       # form feed is not JSON whitespace
    (lambda t: t.replace('"extra_unknown": 7', '"' + "k" * 127 + '": 7'), 0),       # keys up to 127 bytes
    (lambda t: t.replace('"extra_unknown": 7', '"' + "k" * 128 + '": 7'), -2),
    (lambda t: t.replace('"extra_unknown": 7', '"extra_unknown": [7]'), -2),        # nested unknown value
    (lambda t: t.replace('"third": null', '"third": nullx'), -2),
    (lambda t: t.replace('"fs4_shift_enabled": true', '"fs4_shift_enabled": 1'), -2),
]


@pytest.mark.parametrize("i", range(len(GRAMMAR_CASES)))
def test_sidecar_grammar_corner_cases(built, tmp_path, i):
    edit, want = GRAMMAR_CASES[i]
    text = edit(json.dumps(BASE, indent=1))
    q = write_capture(str(tmp_path), text)
    rc, _ = mine(q)
    assert rc == want, (i, rc, want)
    if orc.have_ref():
        assert ref_meta(q)[0] == want, i


def test_data_file_drive_letter_and_backslash_count_as_absolute(built, tmp_path):
    """path_is_absolute (src/io/iq/iq_replay.c:66-78) takes '\\' and 'X:' for absolute on every host."""
    for name in ("c:p.iq", "\\\\p.iq"):
        q = write_capture(str(tmp_path), dict(BASE, data_file=name, data_bytes=0))
        rc, info = mine(q)
        assert rc == 0 and info.data_path == name.encode(), (name, rc, info.data_path)
        if orc.have_ref():
            assert ref_meta(q)[3] == name.encode()


@needs_ref
def test_sidecar_differential_fuzz_vs_reference(built, tmp_path):
    """900 randomly damaged sidecars (characters replaced / deleted / inserted, truncations): this reader and the
    reference's return the same code for metadata-only and for replay opens, and the same fields when both accept."""
    import random
    tmp = str(tmp_path)
    v2 = dict(BASE, version=2, contains_retunes=True, capture_retune_count=1, events=[
        {"kind": "MUTE", "byte_offset": 100, "reason": "squelch", "duration_bytes": 64},
        {"kind": "RETUNE", "byte_offset": 400, "reason": "t", "center_frequency_hz": 852000000,
         "capture_center_frequency_hz": 852000000, "sample_rate_hz": 96000},
        {"kind": "RESET", "byte_offset": 400, "reason": "t", "center_frequency_hz": 852000000,
         "capture_center_frequency_hz": 852000000, "sample_rate_hz": 96000}])
    np.zeros(2001, np.uint8).tofile(os.path.join(tmp, "cap.iq"))
    q = os.path.join(tmp, "cap.iq.json")
    rng = random.Random(11 + 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0")))
    both_ok = 0
    for i in range(900):
        t = list(json.dumps(BASE if i % 3 else v2, indent=1))
        for _ in range(rng.randint(1, 3)):
            op, pos = rng.random(), rng.randrange(len(t))
            if op < 0.35:
                t[pos] = rng.choice('{}[],:"\\0123456789-.eEtfnu \n\t')
            elif op < 0.6:
                del t[pos]
            elif op < 0.8:
                t.insert(pos, rng.choice('{}[],:"\\0 9-'))
            else:
                t = t[:pos]
            if not t:
                t = ["{"]
        with open(q, "w") as f:
            f.write("".join(t))
        for replay in (0, 1):
            rc_r, out, stage, dp, err = ref_meta(q, replay)
            rc_m, info = mine(q, replay)
            assert rc_m == rc_r, (i, replay, rc_m, rc_r, err, "".join(t)[:200])
            if rc_r == 0:
                same_fields(info, out, stage, dp)
                both_ok += 1
    assert both_ok >= 40


def _write_golden_capture(tmp, name, npz, rate=48000, **over):
    g = golden(npz)
    iq = np.ascontiguousarray(g["iq"], np.uint8)
    iq.tofile(os.path.join(tmp, name))
    meta = dict(BASE, sample_rate_hz=rate, base_decimation=1, demod_rate_hz=rate, data_file=name, data_bytes=int(iq.size),
                fs4_shift_enabled=False)
    meta.update(over)
    with open(os.path.join(tmp, name + ".json"), "w") as f:
        json.dump(meta, f)
    return os.path.join(tmp, name), iq


def test_load_batch_shapes_and_mismatch(built, tmp_path):
    tmp = str(tmp_path)
    p1, a = _write_golden_capture(tmp, "cc.iq", "iq_p25p1_c4fm_cc.npz")
    p2, b = _write_golden_capture(tmp, "vc.iq", "iq_p25p1_c4fm_vc.npz")
    l = ddn.lib()
    paths = (C.c_char_p * 2)(p1.encode(), (p2 + ".json").encode())
    buf, n, info = C.c_void_p(), C.c_size_t(), Info()
    assert l.ddn_iq_load_batch(paths, 2, C.byref(buf), C.byref(n), C.byref(info)) == 0
    assert n.value == min(len(a), len(b)) and info.sample_format == 1 and info.demod_rate_hz == 48000
    rows = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), (2, n.value, 2)).copy()
    l.ddn_iq_free(buf)
    assert np.array_equal(rows[0], a[:n.value]) and np.array_equal(rows[1], b[:n.value])
    p3, _ = _write_golden_capture(tmp, "other.iq", "iq_p25p1_c4fm_cc.npz", rate=24000)
    paths = (C.c_char_p * 2)(p1.encode(), p3.encode())
    assert l.ddn_iq_load_batch(paths, 2, C.byref(buf), C.byref(n), None) == -6       # DSD_IQ_ERR_RATE_CHAIN


def test_load_batch_applies_fs4_shift_and_refuses_post_downsample(built, tmp_path):
    """fs4_shift_enabled CU8 captures come out of ddn_iq_load_batch rotated by j^n from the start of the capture: widened,
    they equal the reference's widen_rotate90_u8_to_f32_bias127_phase(phase=0) of the raw bytes bit for bit (numpy
    restatement always; the compiled reference when oracle/_ref is built).  post_downsample != 1 is refused, and a batch
    mixing rotated and unrotated captures is a rate-chain mismatch."""
    tmp = str(tmp_path)
    p1, a = _write_golden_capture(tmp, "rot.iq", "iq_p25p1_c4fm_cc.npz", fs4_shift_enabled=True)
    l = ddn.lib()
    paths = (C.c_char_p * 1)(p1.encode())
    buf, n, info = C.c_void_p(), C.c_size_t(), Info()
    assert l.ddn_iq_load_batch(paths, 1, C.byref(buf), C.byref(n), C.byref(info)) == 0
    assert info.fs4_shift_enabled == 1
    rows = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), (n.value, 2)).copy()
    l.ddn_iq_free(buf)
    raw = a[:n.value].reshape(-1, 2)
    wide = ((rows.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)).astype(np.float32)
    z = ((raw.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)).astype(np.float32)
    ph = np.arange(n.value) & 3
    want = np.empty_like(z)
    want[:, 0] = np.select([ph == 0, ph == 1, ph == 2], [z[:, 0], -z[:, 1], -z[:, 0]], z[:, 1])
    want[:, 1] = np.select([ph == 0, ph == 1, ph == 2], [z[:, 1], z[:, 0], -z[:, 1]], -z[:, 0])
    assert np.array_equal(wide.view(np.uint32), want.view(np.uint32))
    if orc.have_ref():
        r = orc.ref()
        r.widen_rotate90_u8_to_f32_bias127_phase.restype = C.c_uint32
        r.widen_rotate90_u8_to_f32_bias127_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        src = np.ascontiguousarray(raw)
        out = np.zeros(raw.shape, np.float32)
        assert r.widen_rotate90_u8_to_f32_bias127_phase(src.ctypes.data, out.ctypes.data, src.size, 0) == (n.value & 3)
        assert np.array_equal(out.view(np.uint32), wide.view(np.uint32))
    p2, _ = _write_golden_capture(tmp, "plain.iq", "iq_p25p1_c4fm_cc.npz")
    paths = (C.c_char_p * 2)(p1.encode(), p2.encode())
    assert l.ddn_iq_load_batch(paths, 2, C.byref(buf), C.byref(n), None) == -6
    p3, _ = _write_golden_capture(tmp, "pd.iq", "iq_p25p1_c4fm_cc.npz", rate=96000, post_downsample=2, demod_rate_hz=48000)
    paths = (C.c_char_p * 1)(p3.encode())
    assert l.ddn_iq_load_batch(paths, 1, C.byref(buf), C.byref(n), None) == -6
    assert b"post_downsample" in l.ddn_last_error()


@pytest.mark.gpu
def test_capture_files_through_the_chain(built, tmp_path):
    """capture file -> ddn_iq_load_batch -> front end -> rx loop -> NID: the control-channel capture's NAC (the
    reference's DECODE_IQ_P25P1_C4FM_CC answer), read from a dsd-neo-iq file pair instead of an in-memory array."""
    from test_real_capture import nids_from_records, decode_nids
    from test_oracle_block import oracle_nid
    tmp = str(tmp_path)
    p1, a = _write_golden_capture(tmp, "cc.iq", "iq_p25p1_c4fm_cc.npz")
    l = ddn.lib()
    paths = (C.c_char_p * 1)(p1.encode())
    buf, n, info = C.c_void_p(), C.c_size_t(), Info()
    assert l.ddn_iq_load_batch(paths, 1, C.byref(buf), C.byref(n), C.byref(info)) == 0
    iq = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), (1, n.value, 2)).copy()
    l.ddn_iq_free(buf)
    disc = ddn.Batch(1, sample_rate_hz=info.demod_rate_hz, block_len=8192).run_host(iq, n.value)
    rec, fl, cnt = ddn.P25Rx(1, lock_symbols=156, use_matched_filter=1).run(disc)
    r4, _ = orc.unpack_records10(rec[0, :cnt[0]])
    out = decode_nids(nids_from_records(r4, fl[0], int(cnt[0])), oracle_nid)[1:]
    want_nac = int(bytes(golden("iq_p25p1_c4fm_cc.npz")["expected_nac_hex"]).decode(), 16)
    assert len(out) >= 24 and np.all(out[:, 0] == 1) and np.all(out[:, 1] == want_nac) and np.all(out[:, 2] == 7)


@pytest.mark.gpu
def test_decode_capture_tool_reports_the_reference_answers(built, tmp_path):
    """tools/decode_capture.py on the reference's two C4FM captures (written out as dsd-neo-iq file pairs): its lines carry
    the facts the reference's full-chain tests assert - "NAC/CC: 140" and "Group Voice Channel User" (tests/CMakeLists.txt
    DECODE_IQ_P25P1_C4FM_CC / _VOICE)."""
    import sys
    sys.path.insert(0, os.path.join(ddn.ROOT, "tools"))
    import decode_capture
    tmp = str(tmp_path)
    p_cc, _ = _write_golden_capture(tmp, "cc.iq", "iq_p25p1_c4fm_cc.npz")
    p_vc, _ = _write_golden_capture(tmp, "vc.iq", "iq_p25p1_c4fm_vc.npz")
    cc = decode_capture.decode(p_cc, lock=336, out=lambda s: None)      # three-block TSDUs, 360 symbols apart
    good = [t for t in cc[1:] if "NAC 140  TSBK" in t and "crc ok" in t]
    assert len(good) >= 20 and not any("CRC ERR" in t for t in cc[1:])
    vc = decode_capture.decode(p_vc, lock=840, out=lambda s: None)
    assert sum("Group Voice Channel User" in t and "LC ok" in t for t in vc) >= 4
    assert sum("ESS ok: ALGID 80 KID 0000" in t for t in vc) >= 4
    assert all("9 IMBE frames (0 flagged)" in t for t in vc if "LDU" in t and "LC ok" in t)
