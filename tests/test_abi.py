"""CPU: the C-ABI library loads and exports every symbol include/step_amd.h declares; the host
(no-GPU) entry points validate their arguments; the Python binding lists the same symbols."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "step_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|size_t)\s+(\w+)\s*\(", src, flags=re.M)
    assert len(names) >= 20
    return names


def test_library_exports_every_declared_symbol():
    from step_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), "libstep_amd.so does not export %s" % name


def test_header_is_plain_c(tmp_path):
    """include/step_amd.h is a C header (the boundary has no C++ or torch types): it compiles as
    strict C99, and the ctypes mirrors of its structs have the sizes the C compiler gives them."""
    import subprocess
    from step_amd import _lib
    structs = ["bh_tables_t", "bh_meas_t", "bh_weights_t", "bh_step_ws_t", "bh_update_t", "mcmc_params_t",
               "nufft_plan_t"]
    structs = [n for n in structs if hasattr(_lib, n)]
    src = tmp_path / "hdr.c"
    src.write_text('#include <stdio.h>\n#include "step_amd.h"\nint main(void) {\n'
                   + "".join('  printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n in structs)
                   + "  return 0;\n}\n")
    exe = tmp_path / "hdr"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                    "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    assert len(sizes) >= 5
    for n, sz in sizes.items():
        assert ctypes.sizeof(getattr(_lib, n)) == sz, n


def test_plain_c_program_links_and_runs(tmp_path):
    """A C99 program (no Python, no torch) links against libstep_amd.so and calls its host entry
    points: the ABI version, a support query, and the MRI row mask of the shipped configuration
    (52 of 192 rows at acc_factor 6 with 24 ACS lines, dynamic_mri.py:9-39)."""
    import subprocess
    from step_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    src = tmp_path / "use.c"
    src.write_text(r"""
#include <stdio.h>
#include "step_amd.h"
int main(void) {
    float m[192];
    int rc = mri_build_mask(6, 0, 192, 1, 24, m), kept = 0, i;
    for (i = 0; i < 192; ++i) kept += m[i] != 0.f;
    printf("%d %d %d %d %d\n", step_abi_version(), rc, kept, bh_sep3_supported(256, 256), bh_sep3_supported(100, 100));
    return 0;
}
""")
    exe = tmp_path / "use"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lstep_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [_lib.lib().step_abi_version(), 0, 52, 1, 0]


def test_binding_covers_header_exactly():
    from step_amd import _lib
    assert set(_lib.SIGNATURES) == set(header_functions())
    assert _lib.lib().step_abi_version() == _lib.ABI_VERSION == 12


def test_host_entry_points_reject_bad_arguments():
    from step_amd import _lib
    L = _lib.lib()
    m = np.zeros((1, 16), np.float32)
    assert L.mri_build_mask(4, 7, 16, 1, 2, _lib.ptr(m)) != 0
    assert b"Unknown mode" in L.step_last_error()
    assert L.mri_build_mask(0, 0, 16, 1, 2, _lib.ptr(m)) != 0
    with pytest.raises(RuntimeError, match="rc=-1"):
        _lib.check(L.mri_build_mask(4, 0, 16, 1, 2, None), "mri_build_mask")
    # a triangle that needs a baseline the table does not hold -> error (reference: KeyError)
    rt = np.array([1.0, 1.0]); a = np.array([0, 0], np.int32); b = np.array([1, 2], np.int32)
    tt = np.array([1.0]); ts = np.array([[0, 1, 2]], np.int32)
    out = [np.zeros((1, 3), np.int64) for _ in range(2)]
    rc = L.bh_build_closure_tables(_lib.ptr(rt), _lib.ptr(a), _lib.ptr(b), 2, _lib.ptr(tt), _lib.ptr(ts), 1,
                                   None, None, 0, _lib.ptr(out[0]), _lib.ptr(out[1]), None, None)
    assert rc != 0 and b"no baseline" in L.step_last_error()
    # GPU entry points validate shapes before touching the device
    assert L.bh_dft_fwd_sep(None, None, 64, 28, 256, 256, None, 1, 1.0, 0.0, 0.0, None, None, None) != 0
    assert L.mri_forward(None, 1, 12, 192, None, None, 1, 52, None, None) != 0
    assert L.bh_dft_fwd_sep3(None, None, 64, 28, 256, 256, None, 1, 1.0, 0.0, 0.0, None, None, None) != 0
    assert [L.bh_sep3_supported(n, n) for n in (32, 64, 96, 128, 256, 512)] == [0, 1, 0, 1, 1, 0]
    assert L.bh_sep3_supported(256, 128) == 0
    # fold precondition: Re ey even / Im ey odd about the mid-line, bit for bit (host check)
    y = (np.arange(8) - 3.5)[None, None, None, :] * np.linspace(0.1, 0.9, 32)[None, None, :, None]
    ey = np.concatenate([np.cos(y), np.sin(y)], axis=1).astype(np.float32)      # [1][2][32][8]
    ey = np.ascontiguousarray(np.repeat(ey, 2, axis=0))
    assert L.bh_sep3_tables_ok(_lib.ptr(ey), 2, 8) == 1
    ey[1, 1, 4, 6] = np.nextafter(ey[1, 1, 4, 6], np.float32(9))
    assert L.bh_sep3_tables_ok(_lib.ptr(ey), 2, 8) == 0
    assert L.bh_sep3_tables_ok(None, 2, 8) == 0
    assert L.bh_dense_rows_padded(28) == 64 and L.bh_dense_rows_padded(6) == 32
    assert L.bh_dense_packed_floats(64, 28, 65536) == 64 * 64 * 65536


def test_missing_library_fails_loudly(monkeypatch):
    from step_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libstep_amd.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_build_entry_point_runs():
    """__graft_entry__.build(): make (a no-op when the tree is built) + load + the ABI version of the bindings."""
    import __graft_entry__ as g
    g.build()
