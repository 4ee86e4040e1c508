"""CPU: the mixed-radix Stockham passes behind the any-resolution MRI kernels (step_amd/csrc/fft_any.h)
are plain C++; compile them with g++ and compare with numpy's FFT, forward and unnormalised inverse, for
lengths with every kind of factor (powers of 4 and 2, 3, larger primes through the direct butterfly, a prime
length), several lines side by side with a padded stride, the in-place register passes simulated
thread by thread (all gathers, then all scatters) as the kernels' barrier orders them."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "fft_any.h"
// argv: N nl LD sign nthreads ; stdin: nl lines of N complex (line-major) ; stdout: the same layout
int main(int argc, char** argv) {
    const int N = atoi(argv[1]), nl = atoi(argv[2]), LD = atoi(argv[3]), sign = atoi(argv[4]), nth = atoi(argv[5]);
    FftAnyPlan plan;
    if (!fft_any_plan(N, &plan)) return 2;
    std::vector<cf32> in((size_t)N * nl), a((size_t)N * LD), b((size_t)N * LD), tw(N);
    if (fread(in.data(), sizeof(cf32), in.size(), stdin) != in.size()) return 1;
    for (int e = 0; e < N; ++e) tw[e] = c_make((float)cos(2 * M_PI * e / N), (float)sin(2 * M_PI * e / N));
    for (int c = 0; c < nl; ++c)
        for (int i = 0; i < N; ++i) a[(size_t)i * LD + c] = in[(size_t)c * N + i];
    // the kernels' driver (any_fft, mri_any.hip): radices 16, 12, 8, 4, 3, 2, 5, 7 in place through per-thread registers
    // (gather by all threads, then scatter by all threads: the barrier in between), others out of place
    int nlp = 1;
    while (nlp < nl) nlp *= 2;
    if (nth % nlp) return 3;
    cf32 *pa = a.data(), *pb = b.data();
    int s = 1;
    for (int i = 0; i < plan.npass; ++i) {
        const int r = plan.radix[i];
#define REG_PASS(R)                                                                              \
        {                                                                                        \
            constexpr int MB = FFT_ANY_REGS / R;                                                 \
            if (N / R > MB * (nth / nlp)) return 4;                                              \
            std::vector<cf32> regs((size_t)nth * MB * R);                                        \
            for (int tid = 0; tid < nth; ++tid)                                                  \
                fft_any_gather<R, MB>(pa, N, nl, nlp, LD, tid, nth, *(cf32(*)[MB][R])&regs[(size_t)tid * MB * R]); \
            for (int tid = 0; tid < nth; ++tid) {                                                \
                if (sign < 0) fft_any_scatter<-1, R, MB>(pa, tw.data(), N, s, plan.magic[i], nl, nlp, LD, tid, nth, *(cf32(*)[MB][R])&regs[(size_t)tid * MB * R]); \
                else fft_any_scatter<1, R, MB>(pa, tw.data(), N, s, plan.magic[i], nl, nlp, LD, tid, nth, *(cf32(*)[MB][R])&regs[(size_t)tid * MB * R]); \
            }                                                                                    \
        }
        if (r == 16) REG_PASS(16)
        else if (r == 12) REG_PASS(12)
        else if (r == 8) REG_PASS(8)
        else if (r == 2) REG_PASS(2)
        else if (r == 3) REG_PASS(3)
        else if (r == 4) REG_PASS(4)
        else if (r == 5) REG_PASS(5)
        else if (r == 7) REG_PASS(7)
        else {
            for (int tid = 0; tid < nth; ++tid) {
                if (sign < 0) fft_any_pass<-1>(pa, pb, tw.data(), N, s, r, nl, LD, tid, nth);
                else fft_any_pass<1>(pa, pb, tw.data(), N, s, r, nl, LD, tid, nth);
            }
            for (size_t e = 0; e < (size_t)N * LD; ++e) pa[e] = pb[e];
        }
        s *= r;
    }
    for (int c = 0; c < nl; ++c)
        for (int i = 0; i < N; ++i) in[(size_t)c * N + i] = pa[(size_t)i * LD + c];
    fwrite(in.data(), sizeof(cf32), in.size(), stdout);
    fprintf(stderr, "passes:");
    for (int i = 0; i < plan.npass; ++i) fprintf(stderr, " %d", plan.radix[i]);
    fprintf(stderr, "\n");
    return 0;
}
"""


@pytest.fixture(scope="module")
def any_binary(tmp_path_factory):
    d = tmp_path_factory.mktemp("fft_any")
    src = d / "drv.cpp"
    src.write_text(SRC)
    exe = d / "drv"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "step_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 24, 30, 32, 48, 49, 64, 72, 96, 97, 128, 144, 160, 176, 192, 200, 208, 210, 256, 320, 384,
                               486, 512, 1000, 1024])
def test_any_length_matches_numpy(any_binary, n):
    rng = np.random.RandomState(n)
    nl, ld = 5, 9
    x = (rng.randn(nl, n) + 1j * rng.randn(nl, n)).astype(np.complex64)
    for sign, nth in ((-1, 512 if n <= 512 else 2048), (1, 1024 if n <= 512 else 4096)):
        r = subprocess.run([any_binary, str(n), str(nl), str(ld), str(sign), str(nth)], input=x.tobytes(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        got = np.frombuffer(r.stdout, np.complex64).reshape(nl, n)
        x64 = x.astype(np.complex128)
        want = np.fft.fft(x64, axis=1) if sign < 0 else np.fft.ifft(x64, axis=1) * n
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err <= 1e-6, "N=%d sign=%d (%s): relative error %.2e" % (n, sign, r.stderr.decode().strip(), err)


def test_plan_refuses_what_it_cannot_hold(any_binary):
    r = subprocess.run([any_binary, "1025", "1", "1", "-1", "1"], input=b"", stdout=subprocess.PIPE)
    assert r.returncode == 2


PLAN_SRC = r"""
#include <cstdio>
#include "fft_any.h"
// every length 2 .. 1024: a plan exists, its radices multiply to N, and the kernels' line counts (512 threads,
// 64 KB of line buffers, with and without the padded stride) are positive with everything inside 160 KB of LDS
int main() {
    int bad = 0;
    for (int n = 2; n <= FFT_ANY_MAX_N; ++n) {
        FftAnyPlan p;
        if (!fft_any_plan(n, &p)) { printf("no plan %d\n", n); ++bad; continue; }
        long prod = 1;
        for (int i = 0; i < p.npass; ++i) prod *= p.radix[i];
        if (prod != n) { printf("radices of %d multiply to %ld\n", n, prod); ++bad; }
        for (int pad = 0; pad < 2; ++pad) {
            const int nl = fft_any_lines(n, &p, 512, pad, 64 * 1024);
            const size_t lds = ((size_t)n + (size_t)(p.direct ? 2 : 1) * n * (nl + pad) + nl) * sizeof(cf32) + 64;
            if (nl < 1 || lds > 160 * 1024) { printf("n %d pad %d: %d lines, %zu bytes\n", n, pad, nl, lds); ++bad; }
            for (int i = 0; i < p.npass; ++i)
                if (fft_any_in_regs(p.radix[i]) && n / p.radix[i] > fft_any_maxb(p.radix[i]) * (512 / nl)) {
                    printf("n %d radix %d does not fit the registers at %d lines\n", n, p.radix[i], nl); ++bad;
                }
        }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
"""


def test_every_length_has_a_plan_that_fits(tmp_path):
    src = tmp_path / "plan.cpp"
    src.write_text(PLAN_SRC)
    exe = tmp_path / "plan"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "step_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("bad 0"), r.stdout[-2000:]
