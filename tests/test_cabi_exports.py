"""CPU: libdsdneo_hip.so loads without a GPU and exports every symbol include/*.h declares; compute calls
fail loudly (DDN_ENODEV) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import pytest

import ddn


def declared_functions():
    text = "".join(open(os.path.join(ddn.ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ddn.ROOT, "include"))))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_0-9]+)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith(("ddn_", "simd_", "widen_", "p25_", "dmr_", "viterbi_",
                                                         "CNXDN", "check_", "hamming_", "golay_", "bch_", "p25p1_", "dsd_", "crc16_", "mbe_", "Hamming_", "Golay_", "QR_", "BPTC", "rs_12_9_", "InitAll", "trellis_", "full_demod", "op25_", "ez_rs28_", "isch_", "p25p2_"))))


def test_header_and_binding_agree(built):
    assert declared_functions() == sorted(ddn.PROTOTYPES.keys())


def test_library_exports_every_declared_symbol(built):
    l = C.CDLL(ddn.LIB_PATH)
    for name in declared_functions():
        assert hasattr(l, name), name
    assert ddn.lib().ddn_version().startswith(b"dsdneo-hip")
    assert ddn.lib().simd_fir_get_impl_name() == b"hip-gfx950"


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ddn.DdnError) as e:
        ddn.Batch(4)
    assert "rc=-2" in str(e.value)


def test_product_library_reads_no_environment(built):
    """the experiment knobs of the timing tools (DDN_EXP_ENV, dsd-neo_amd/csrc/ddn_device.h) compile to nothing in the product build:
    libdsdneo_hip.so does not import getenv at all, and the sources call it nowhere but in that one macro"""
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", ddn.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert not re.search(r"\b(secure_)?getenv\b", und), [l for l in und.splitlines() if "getenv" in l]
    src = os.path.join(ddn.ROOT, "dsd-neo_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        for k, line in enumerate(open(os.path.join(src, f), errors="replace")):
            if re.search(r"\bgetenv\s*\(", line):
                hits.append((f, k + 1))
    assert hits == [("ddn_device.h", hits[0][1])] if hits else False, hits


def test_product_does_not_reference_oracle():
    """The shipped sources never include/link anything under oracle/."""
    for sub in ("dsd-neo_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ddn.ROOT, sub)):
            if "build" in dp:
                continue
            for fn in fns:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                t = open(os.path.join(dp, fn), errors="ignore").read()
                assert "ddn_oracle" not in t and "oracle/" not in t.replace("oracle/_ref", ""), os.path.join(dp, fn)


def test_c_host_example_compiles_and_links(built, tmp_path):
    """INTEGRATION.md's C host over ddn_p25_chain (examples/p25_chain_host.c) builds with plain gcc against include/ and the
    library - the boundary is C, no HIP or torch on the caller's side; without a GPU it fails loudly at create"""
    import subprocess
    exe = str(tmp_path / "p25_chain_host")
    lib_dir = os.path.dirname(ddn.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ddn.ROOT, "include"),
                           os.path.join(ddn.ROOT, "examples", "p25_chain_host.c"), "-L", lib_dir, "-ldsdneo_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    exe2 = str(tmp_path / "p25_node_host")                  # the same for every GPU of the node (include/ddn_node.h)
    subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Werror", "-I", os.path.join(ddn.ROOT, "include"),
                           os.path.join(ddn.ROOT, "examples", "p25_node_host.c"), "-L", lib_dir, "-ldsdneo_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe2])
    exe3 = str(tmp_path / "mixed_node_host")                # BASELINE configs[3] from C: kind = DDN_NODE_MIXED
    subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Werror", "-I", os.path.join(ddn.ROOT, "include"),
                           os.path.join(ddn.ROOT, "examples", "mixed_node_host.c"), "-L", lib_dir, "-ldsdneo_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe3])
    import torch
    if not torch.cuda.is_available():
        for e in (exe, exe2, exe3):
            p = subprocess.run([e], capture_output=True, text=True, timeout=120)
            assert p.returncode == 1 and p.stderr.strip()   # ddn_last_error(): no device - never a CPU fallback


def test_configure_probe_of_the_reference_links_and_unsupported_rates_say_so(built, tmp_path):
    """dsd-neo's configure step links a probe that names every mbelib-neo symbol it uses (CMakeLists.txt:626-657); the two rates
    that are not restated here (IMBE 7100x4400, AMBE 2400 data) are present, decode nothing and report MBE_STATUS_UNSUPPORTED"""
    import ctypes as C
    import subprocess
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "probe.c"
    src.write_text('''
#include "ddn_mbe.h"
int main(void) {
    char imbe_fr[8][23] = {{0}}, ambe_fr[4][24] = {{0}}, imbe7100_fr[7][24] = {{0}}, imbe_d[88] = {0}, ambe_d[49] = {0}, str[64];
    float audio[160];
    mbe_parms cur, prev, prev_enhanced;
    mbe_process_result result;
    mbe_initMbeParms(&cur, &prev, &prev_enhanced);
    mbe_initProcessResult(&result);
    (void)mbe_decodeImbe7200x4400Frame(imbe_fr, imbe_d, &result);
    (void)mbe_decodeAmbe3600x2450Frame(ambe_fr, ambe_d, &result);
    (void)mbe_decodeImbe7100x4400Frame(imbe7100_fr, imbe_d, &result);
    (void)mbe_processImbe4400Dataf(audio, &result, imbe_d, &cur, &prev, &prev_enhanced);
    (void)mbe_processAmbe2450Dataf(audio, &result, ambe_d, &cur, &prev, &prev_enhanced);
    (void)mbe_processAmbe2400Dataf(audio, &result, ambe_d, &cur, &prev, &prev_enhanced);
    mbe_formatProcessResult(str, sizeof str, &result);
    mbe_synthesizeSilencef(audio);
    return 0;
}
''')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", os.path.join(root, "dsd-neo_amd"),
                    "-ldsdneo_hip", "-Wl,-rpath," + os.path.join(root, "dsd-neo_amd")], check=True)
    l = ddn.lib()
    res = ddn.MbeProcessResult()
    fr, d = np.ones((7, 24), np.uint8), np.ones(88, np.uint8)
    assert l.mbe_decodeImbe7100x4400Frame(fr.ctypes.data, d.ctypes.data, C.byref(res)) == -4 and not d.any() and res.flags == 0x10
    pcm, ad = np.ones(160, np.float32), np.zeros(49, np.uint8)
    assert l.mbe_processAmbe2400Dataf(pcm.ctypes.data, C.byref(res), ad.ctypes.data, None, None, None) == -4
    assert not pcm.any() and res.flags == 0x10


def test_every_mbelib_symbol_the_reference_calls_is_exported(built):
    """tests/golden/mbe_symbols_called.json (tools/gen_mbe_symbols.py) lists every mbe_* function the reference's src/ and include/
    call without defining it - the link-time contract of B6, wider than the configure probe.  Each one is an exported symbol of the
    library; the three outside the probe behave as include/ddn_mbe.h says"""
    import ctypes as C
    import json
    import numpy as np
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mbe_symbols_called.json")))["symbols"]
    assert len(want) >= 15
    l = ddn.lib()
    missing = [name for name in want if not hasattr(l, name)]
    assert not missing, missing
    res = ddn.MbeProcessResult()
    fr, d, pcm = np.ones((4, 24), np.uint8), np.ones(49, np.uint8), np.ones(160, np.float32)
    assert l.mbe_processAmbe3600x2400Framef(pcm.ctypes.data, C.byref(res), fr.ctypes.data, d.ctypes.data, None, None, None) == -4
    assert not pcm.any() and not d.any() and res.flags == 0x10
    f = np.array([0.0, 1.0, -1.0, 0.99, -0.5, 5000.0, -5000.0, 4680.1] + [0.25] * 152, np.float32)
    sh = np.zeros(160, np.int16)
    l.mbe_floattoshort(f.ctypes.data, sh.ctypes.data)
    assert sh[:8].tolist() == [0, 7, -7, 6, -3, 32760, -32760, 32760] and (sh[8:] == 1).all()
    l.mbe_versionString.restype = C.c_char_p
    assert b"mbelib-neo 2.0" in l.mbe_versionString()


def test_node_configuration_is_checked_before_any_device_is_touched(built):
    """ddn_node_create (include/ddn_node.h): a configuration that names no chain kind, a kind without its settings, or no channels is
    DDN_EINVAL with a message - on a box without a GPU too; a good one without a device is DDN_ENODEV (never a CPU path)"""
    import ctypes as C
    import torch
    l = ddn.lib()
    h = C.c_void_p()

    def create(**kw):
        base = dict(n_channels=8, samples_per_call=4096, block_len=4096, input_format=0, vocoder=0, modulation=0, n_devices=1, kind=0,
                    n_dmr=0, n_nxdn48=0, overlap=0, fsk4=None, p25p2=None, p25p2_seed44=None)
        base.update(kw)
        cfg = ddn.NodeConfig(*[base[k] for k, _ in ddn.NodeConfig._fields_])
        return l.ddn_node_create(C.byref(cfg), C.byref(h))

    assert create(kind=7) == -1 and b"kind" in l.ddn_last_error()
    assert create(kind=ddn.NODE_FSK4) == -1            # no ddn_fsk4_chain_config
    assert create(kind=ddn.NODE_P25P2) == -1           # no ddn_p25p2_chain_config
    assert create(n_channels=0) == -1
    assert create(kind=ddn.NODE_MIXED, n_channels=0, n_dmr=-1) == -1
    assert create(samples_per_call=0) == -1
    assert l.ddn_node_kind_of(None) == -1 and l.ddn_node_chain_object(None, 0) is None
    assert l.ddn_node_on_part(None, 0, None, None) == -1
    if not torch.cuda.is_available():
        assert create() == -2 and l.ddn_last_error()
        assert create(kind=ddn.NODE_MIXED, n_dmr=4, n_nxdn48=4) == -2
