"""CPU: the in-register FFT building blocks of the MRI kernels (step_amd/csrc/fft_small.h: 3-, 4-,
12-, 16-, 24-point transforms and the 192 = 16 x 12 / 384 = 16 x 24 two-pass decompositions) are plain C++; compile them
with g++ and compare with numpy's FFT, forward and unnormalised inverse."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cmath>
#include "fft_small.h"
// stdin-free driver: reads N complex numbers per case from argv-less binary stdin
template <int SIGN> void fft192(const cf32* in, cf32* out) {
    cf32 tw[192];
    for (int m = 0; m < 192; ++m) tw[m] = c_make((float)cos(2 * M_PI * m / 192), (float)sin(2 * M_PI * m / 192));
    cf32 slot[192];
    for (int b = 0; b < 12; ++b) {                 // pass A: task b, inputs x[12 a + b]
        cf32 v[16];
        for (int a = 0; a < 16; ++a) v[a] = in[12 * a + b];
        fft192_pass_a<SIGN>(v, b, tw);
        for (int c = 0; c < 16; ++c) slot[12 * c + b] = v[c];
    }
    for (int c = 0; c < 16; ++c) {                 // pass B: task c, outputs X[c + 16 d]
        cf32 v[12];
        for (int b = 0; b < 12; ++b) v[b] = slot[12 * c + b];
        fft192_pass_b<SIGN>(v);
        for (int d = 0; d < 12; ++d) out[c + 16 * d] = v[d];
    }
}
template <int SIGN> void fft384(const cf32* in, cf32* out) {
    static cf32 tw[384], slot[384];
    for (int m = 0; m < 384; ++m) tw[m] = c_make((float)cos(2 * M_PI * m / 384), (float)sin(2 * M_PI * m / 384));
    for (int b = 0; b < 24; ++b) {
        cf32 v[16];
        for (int a = 0; a < 16; ++a) v[a] = in[24 * a + b];
        fft384_pass_a<SIGN>(v, b, tw);
        for (int c = 0; c < 16; ++c) slot[24 * c + b] = v[c];
    }
    for (int c = 0; c < 16; ++c) {
        cf32 v[24];
        for (int b = 0; b < 24; ++b) v[b] = slot[24 * c + b];
        fft384_pass_b<SIGN>(v);
        for (int d = 0; d < 24; ++d) out[c + 16 * d] = v[d];
    }
}
int main() {
    static cf32 in[384], out[384];
    if (fread(in, sizeof(cf32), 384, stdin) != 384) return 1;
    { cf32 v[16]; for (int i = 0; i < 16; ++i) v[i] = in[i]; fft16<-1>(v); fwrite(v, sizeof(cf32), 16, stdout); }
    { cf32 v[16]; for (int i = 0; i < 16; ++i) v[i] = in[i]; fft16<1>(v); fwrite(v, sizeof(cf32), 16, stdout); }
    { cf32 v[12]; for (int i = 0; i < 12; ++i) v[i] = in[i]; fft12<-1>(v); fwrite(v, sizeof(cf32), 12, stdout); }
    { cf32 v[12]; for (int i = 0; i < 12; ++i) v[i] = in[i]; fft12<1>(v); fwrite(v, sizeof(cf32), 12, stdout); }
    fft192<-1>(in, out); fwrite(out, sizeof(cf32), 192, stdout);
    fft192<1>(in, out); fwrite(out, sizeof(cf32), 192, stdout);
    { cf32 v[24]; for (int i = 0; i < 24; ++i) v[i] = in[i]; fft24<-1>(v); fwrite(v, sizeof(cf32), 24, stdout); }
    { cf32 v[24]; for (int i = 0; i < 24; ++i) v[i] = in[i]; fft24<1>(v); fwrite(v, sizeof(cf32), 24, stdout); }
    fft384<-1>(in, out); fwrite(out, sizeof(cf32), 384, stdout);
    fft384<1>(in, out); fwrite(out, sizeof(cf32), 384, stdout);
    return 0;
}
"""


@pytest.fixture(scope="module")
def fft_binary(tmp_path_factory):
    d = tmp_path_factory.mktemp("fft_small")
    src = d / "drv.cpp"
    src.write_text(SRC)
    exe = d / "drv"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "step_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("seed", [0, 1])
def test_small_ffts_match_numpy(fft_binary, seed):
    rng = np.random.RandomState(seed)
    x = (rng.randn(384) + 1j * rng.randn(384)).astype(np.complex64)
    out = subprocess.run([fft_binary], input=x.tobytes(), stdout=subprocess.PIPE, check=True).stdout
    got = np.frombuffer(out, np.complex64)
    assert got.size == 16 * 2 + 12 * 2 + 192 * 2 + 24 * 2 + 384 * 2
    x64 = x.astype(np.complex128)
    cases = [("fft16 fwd", got[0:16], np.fft.fft(x64[:16])),
             ("fft16 inv", got[16:32], np.fft.ifft(x64[:16]) * 16),
             ("fft12 fwd", got[32:44], np.fft.fft(x64[:12])),
             ("fft12 inv", got[44:56], np.fft.ifft(x64[:12]) * 12),
             ("fft192 fwd", got[56:248], np.fft.fft(x64[:192])),
             ("fft192 inv", got[248:440], np.fft.ifft(x64[:192]) * 192),
             ("fft24 fwd", got[440:464], np.fft.fft(x64[:24])),
             ("fft24 inv", got[464:488], np.fft.ifft(x64[:24]) * 24),
             ("fft384 fwd", got[488:872], np.fft.fft(x64)),
             ("fft384 inv", got[872:1256], np.fft.ifft(x64) * 384)]
    for name, g, want in cases:
        err = np.abs(g - want).max() / np.abs(want).max()
        assert err <= 5e-7, "%s: relative error %.2e" % (name, err)
