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
                                                         "CNXDN", "check_", "hamming_", "golay_", "bch_", "p25p1_", "dsd_", "crc16_", "mbe_", "Hamming_", "Golay_", "QR_", "BPTC", "rs_12_9_", "InitAll", "trellis_", "full_demod", "op25_", "ez_rs28_", "isch_"))))


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
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert p.returncode == 1 and p.stderr.strip()       # ddn_last_error(): no device - never a CPU fallback
