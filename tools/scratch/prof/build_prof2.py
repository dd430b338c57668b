"""libdsdneo_hip_prof.so: k_p25_rxw with cycle-counter marks around the sections of the fast trip path."""
import os
import subprocess

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
s = open(os.path.join(ROOT, "dsd-neo_amd/csrc/ddn_rx.hip")).read()
i = s.index("k_p25_rxw(const float")
head, body = s[:i], s[i:]


def ins(anchor, code, after=False):
    global body
    assert body.count(anchor) >= 1, anchor
    body = body.replace(anchor, (anchor + code) if after else (code + anchor), 1)


T = "{ const long long t_=__builtin_readcyclecounter(); prof[%d] += t_ - tlast; tlast = t_; }\n"
ins("    const float* rrow = &L.raw[ln][0];",
    "    long long prof2[8] = {0,0,0,0,0,0,0,0}; long long prof[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; long long tlast = __builtin_readcyclecounter(); long long ntrip=0, nfast=0, ngen=0;\n")
ins("                // ---- trip classification ----", T % 0 + "ntrip++;\n")
ins("                if (__any(do_a || do_b)) {\n", T % 1 + "nfast++;\n", after=True)
ins("                    // the five window samples (indices centre - 2", T % 2)
ins("                    if (fab) {\n                        const float sym = acc / 5.0f;", T % 3)
ins("                        commit_pre(sym);\n                        int fl = 0;\n                        float q_max = 0.0f, q_min = 0.0f;\n                        if (do_a) {", T % 4)
ins("                        emit(sym, fl, q_max, q_min);\n                    }\n                }\n                if (!__any(gneed)) {", "")
body = body.replace("                        emit(sym, fl, q_max, q_min);\n                    }\n                }\n                if (!__any(gneed)) {",
                    "                        " + (T % 5).strip() + "\n                        emit(sym, fl, q_max, q_min);\n                    }\n                    " + (T % 6).strip() + "\n                }\n                if (!__any(gneed)) {", 1)
ins("                // ---- symbol start ----", T % 7 + "ngen++;\n")
U = "{ const long long t_=__builtin_readcyclecounter(); prof2[%d] += t_ - tlast; tlast = t_; }\n"
ins("        const float lo = (m1 + m2) * 0.5f, hi = (x1 + x2) * 0.5f;", U % 0)
ins("        s.min = (float)(s.min_sum / (double)MS);", U % 1)
ins("        fl = 1 | (neg ? 4 : 0);", U % 2)
ins("        L.lb[s.lidx][ln] = sym;", U % 3)
ins("        s.lidx = (s.lidx == 23) ? 0 : s.lidx + 1;", U % 4)
ins("        if (s.hist_count >= 8) {", U % 5)
ins("                gblocked = gblocked || blocked;", T % 8)
ins("    if (loader && offload && it > 0) {\n        drain((it - 1) & 1);\n    }",
    '    if (!loader && blockIdx.x == 7 && lane == 0) { printf("PROF trips %lld fast %lld generic %lld | loop/other %lld classify %lld search %lld '
    'window %lld div+pre %lld commit %lld emit %lld tail %lld generic %lld\\n", ntrip, nfast, ngen, prof[0], prof[1], prof[2], prof[3], '
    "prof[4], prof[5], prof[6], prof[7], prof[8]); printf(\"PROF2 pre-winpush %lld winpush_if %lld ring+sums %lld thresholds %lld | hunt-pre %lld winpush_h %lld idx+hist %lld\\n\", prof2[0]-0, prof2[0], prof2[1], prof2[2], prof2[3], prof2[4], prof2[5]); }\n")
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
