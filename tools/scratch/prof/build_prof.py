"""Builds tools/scratch/prof/libdsdneo_hip_prof.so: libdsdneo_hip.so with k_p25_rxw instrumented by readcyclecounter marks
(one counter per section of the per-trip loop, printed by lane 0 of workgroup 7)."""
import os
import subprocess

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
s = open(os.path.join(ROOT, "dsd-neo_amd/csrc/ddn_rx.hip")).read()
i = s.index("k_p25_rxw(const float")
head, body = s[:i], s[i:]


def ins(anchor, code):
    global body
    assert body.count(anchor) >= 1, anchor
    body = body.replace(anchor, code + anchor, 1)


T = "{ const long long t_=__builtin_readcyclecounter(); prof[%d] += t_ - tlast; tlast = t_; }\n"
ins("    const float* rrow = &L.raw[ln][0];",
    "    long long prof[10] = {0,0,0,0,0,0,0,0,0,0}; long long tlast = __builtin_readcyclecounter(); long long ntrip=0;\n")
ins("                // ---- symbol start ----", T % 0 + "ntrip++;\n")
ins("                // ---- whole-symbol evaluation ----", T % 1)
ins("                if (latched) {", T % 2)
ins("                const bool genf = wholeok", T % 3)
ins("                const bool gen = wholeok && fits && !latched && !genf;", T % 4)
ins("                // ---- sample-at-a-time path", T % 5)
ins("                // ---- symbol commit ----", T % 6)
ins("                    if (offload && qk < QCW) {", T % 7)
ins("                const bool busy = live && sp < tn && !blocked;", T % 8)
ins("    if (loader && offload && it > 0) {\n        drain((it - 1) & 1);\n    }",
    '    if (!loader && blockIdx.x == 7 && lane == 0) { printf("PROF trips %lld  wait/other %lld start %lld whole %lld '
    'latched %lld genf %lld gen %lld slow %lld commit %lld queue %lld\\n", ntrip, prof[0], prof[1], prof[2], prof[3], '
    "prof[4], prof[5], prof[6], prof[7], prof[8]); }\n")
here = os.path.dirname(os.path.abspath(__file__))
open(os.path.join(here, "ddn_rx_prof.hip"), "w").write(head + body)
amd = os.path.join(ROOT, "dsd-neo_amd")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()
subprocess.check_call(["hipcc"] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(amd, "csrc"), "-c",
                       os.path.join(here, "ddn_rx_prof.hip"), "-o", "/tmp/rx_prof.o"])
objs = [os.path.join(amd, "build", f) for f in os.listdir(os.path.join(amd, "build")) if f.endswith(".o") and f != "ddn_rx.hip.o"]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(here, "libdsdneo_hip_prof.so")]
                      + objs + ["/tmp/rx_prof.o", "-lm"])
print("built")
