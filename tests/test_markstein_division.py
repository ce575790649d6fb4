"""CPU: the three-instruction division of dwt97_l0wg.inc -- q0 = v * r, q = fma(fma(-q0, s, v), r, q0) with r = RN(1 / s) -- gives
the correctly rounded v / s (what the reference's `v / step`, encoder.go:269-274, computes) for every step = 1.0 / Quality the
plan sends down that path (Quality 1..8191) on 10^7 coefficient-like values: compared with the C compiler's IEEE division."""
import ctypes
import os
import subprocess
import tempfile

SRC = r"""
#include <math.h>
#include <stdint.h>
static uint64_t s_ = 88172645463325252ull;
static inline uint64_t rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }
long check(int q_lo, int q_hi, long per_q) {
    long bad = 0;
    for (int Q = q_lo; Q <= q_hi; Q++) {
        const double s = 1.0 / (double)Q, r = 1.0 / s;
        for (long i = 0; i < per_q; i++) {
            const uint64_t u = rnd();
            /* coefficient-like: +-(mantissa) * 2^e, e in [-40, 24]; plus exact small integers and halves now and then */
            double v = ldexp((double)(int64_t)(u >> 11) / 9007199254740992.0, (int)((u >> 3) % 65) - 40);
            if ((u & 7) == 0) v = (double)((int64_t)(u >> 40) - (1 << 23)) * 0.5;
            if (u & 4) v = -v;
            const double q0 = v * r;
            const double q = fma(fma(-q0, s, v), r, q0);
            if (q != v / s) bad++;
        }
    }
    return bad;
}
"""


def test_markstein_division_matches_ieee_division():
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "m.c"); so = os.path.join(d, "m.so")
        open(c, "w").write(SRC)
        hw = "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
        # (without the instruction libm's fma() is the software one: slower, equally exact)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off"] + (["-mfma"] if hw else []) + ["-shared", "-fPIC", "-o", so, c, "-lm"])
        L = ctypes.CDLL(so)
        L.check.restype = ctypes.c_long
        L.check.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
        assert L.check(1, 100, 100000) == 0            # every Quality an Options.Quality in 1..100 can be: 10^7 values
        assert L.check(101, 8191, 1500) == 0           # the rest of the range the plan admits: 1.2 * 10^7 values
