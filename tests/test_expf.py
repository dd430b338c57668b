"""dsd-neo_amd/csrc/ddn_expf.h (the binary32 exp of the M17 soft costs, evaluated as glibc's expf evaluates it) compiled for the host
against this machine's libm expf: every binary32 of a stride through |x| <= 17, plus the neighbourhoods of the clamp and of zero.
(The exhaustive run over all 2.2e9 values takes half a minute: DDN_EXPF_FULL=1.)"""
import os
import subprocess
import tempfile

SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ddn_expf.h"
int main(int argc, char** argv) {
    const uint32_t step = (uint32_t)strtoul(argv[1], NULL, 10);
    unsigned long long bad = 0, n = 0;
    for (int sign = 0; sign < 2; sign++) {
        for (uint64_t b = 0; b <= 0x41880000u; b += step) {
            const uint32_t u = (uint32_t)b | ((uint32_t)sign << 31);
            float x; memcpy(&x, &u, 4);
            const float a = expf(x), c = ddn_expf(x);
            bad += memcmp(&a, &c, 4) != 0;
            n++;
        }
    }
    for (int k = -2000; k <= 2000; k++) {              /* around +-16 (the caller's clamp) and +-1 */
        const float xs[4] = {16.0f, -16.0f, 1.0f, -1.0f};
        for (int q = 0; q < 4; q++) {
            uint32_t u; memcpy(&u, &xs[q], 4); u += (uint32_t)k;
            float x; memcpy(&x, &u, 4);
            const float a = expf(x), c = ddn_expf(x);
            bad += memcmp(&a, &c, 4) != 0;
            n++;
        }
    }
    printf("%llu %llu\n", n, bad);
    return 0;
}
'''


def test_expf_equals_libm():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "dsd-neo_amd", "csrc"), os.path.join(d, "t.c"), "-o", exe, "-lm"])
        step = "1" if os.environ.get("DDN_EXPF_FULL") else "97"
        n, bad = (int(v) for v in subprocess.check_output([exe, step]).split())
    assert n > 2e7 and bad == 0, (n, bad)
