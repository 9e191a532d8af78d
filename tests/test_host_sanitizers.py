"""The HOST-ONLY builders (recad_amd/csrc/host/*.h: the LDS plan builder with its std::thread pool, the CSR schedule builder)
under AddressSanitizer + UBSan and under ThreadSanitizer (CPU builds only): `make -C recad_amd/csrc host-asan
host-tsan` compiles tests/tools/host_builders_driver.cpp, which runs both builders on a stress set of graphs from several
caller threads at once.  Both runs must be clean AND produce the product library's words bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from recad_amd import _lib, synth
from tests.test_lds_plan_cpu import norm_adj_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recad_amd", "csrc")


def _graphs():
    """(U, I, dim, n_cu, split, rowptr, col, val): the reference-shaped graphs + the randomised family of
    tests/tools/spmm_lds_stress.py (tiny classes, empty rows, full rows, dense and sparse; every dim % 4 == 0 up to 256)."""
    out = []
    for shape, dims in (("tiny", (16, 32, 64, 128, 256)), ("ml1m", (64,))):
        data = synth.make(shape)
        U, I = data["n_users"], data["n_items"]
        rp, col, val = norm_adj_csr(U, I, *data["train"])
        for d in dims:
            out.append((U, I, d, 256, 1, rp, col, val))
    rng = np.random.default_rng(7)
    for case in range(40):
        U = int(rng.choice([1, 3, 17, 64, 300, 1500, 5000]))
        I = int(rng.choice([1, 2, 16, 33, 200, 1200, 3700]))
        d = int(rng.choice([4, 8, 12, 32, 48, 64, 100, 128, 256]))
        dens = float(rng.choice([0.002, 0.02, 0.2, 0.9]))
        deg = rng.binomial(I, dens, U)
        if rng.random() < 0.5:
            deg[rng.integers(0, U)] = I
        if rng.random() < 0.5:
            deg[rng.integers(0, U, max(1, U // 10))] = 0
        if deg.sum() == 0:
            deg[0] = min(I, 1)
        ptr = np.zeros(U + 1, dtype=np.int64)
        ptr[1:] = np.cumsum(deg)
        idx = np.concatenate([np.sort(rng.choice(I, size=int(k), replace=False)) for k in deg]).astype(np.int32)
        rp, col, val = norm_adj_csr(U, I, ptr, idx)
        out.append((U, I, d, int(rng.choice([8, 64, 256])), int(rng.integers(0, 2)), rp, col, val))
    return out


def _product_words(g):
    """the shipped library's builders on the same graph (pure host entry points: no GPU needed)"""
    U, I, d, n_cu, split, rp, col, val = g
    L = _lib.lib()
    rp_, col_, val_ = (np.ascontiguousarray(a) for a in (rp.astype(np.int32), col.astype(np.int32), val.astype(np.float32)))
    plan, n_words, info = C.c_void_p(), C.c_int64(0), _lib.LdsInfo()
    _lib.check(L.rk_lds_plan_build_host(U, I, rp_.ctypes.data_as(C.c_void_p), col_.ctypes.data_as(C.c_void_p), val_.ctypes.data_as(C.c_void_p),
                                        d, n_cu, C.byref(plan), C.byref(n_words), C.byref(info)), "rk_lds_plan_build_host")
    pw = np.zeros(n_words.value, dtype=np.int32)
    if n_words.value:
        _lib.check(L.rk_lds_plan_words(plan, pw.ctypes.data_as(C.c_void_p)), "rk_lds_plan_words")
        L.rk_lds_plan_destroy(plan)
    sched, n_blocks, sw, scr = C.c_void_p(), C.c_int32(0), C.c_int64(0), C.c_int64(0)
    _lib.check(L.rk_csr_schedule_build_host(U + I, rp_.ctypes.data_as(C.c_void_p), U if split else 0, d, C.byref(sched), C.byref(n_blocks),
                                            C.byref(sw), C.byref(scr)), "rk_csr_schedule_build_host")
    words = np.zeros(sw.value, dtype=np.int32)
    _lib.check(L.rk_csr_schedule_words(sched, words.ctypes.data_as(C.c_void_p)), "rk_csr_schedule_words")
    L.rk_csr_schedule_destroy(sched)
    return pw, words, n_blocks.value, scr.value


@pytest.fixture(scope="module")
def stress_set(tmp_path_factory):
    gs = _graphs()
    path = str(tmp_path_factory.mktemp("hostsan") / "graphs.bin")
    with open(path, "wb") as f:
        f.write(np.int32(len(gs)).tobytes())
        for U, I, d, n_cu, split, rp, col, val in gs:
            f.write(np.asarray([U, I, d, n_cu, split, len(col)], dtype=np.int32).tobytes())
            f.write(rp.astype(np.int32).tobytes()); f.write(col.astype(np.int32).tobytes()); f.write(val.astype(np.float32).tobytes())
    return gs, path, [_product_words(g) for g in gs]


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_builders_under_sanitizers(san, stress_set, tmp_path):
    gs, path, want = stress_set
    exe = os.path.join(ROOT, "recad_amd", "lib", f"host_builders_{san}")
    mk = subprocess.run(["make", "-C", CSRC, f"host-{san}"], capture_output=True, text=True)
    if mk.returncode != 0 or not os.path.exists(exe):
        pytest.skip(f"the host toolchain cannot build the {san} driver here: {mk.stderr[-300:]}")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    out = str(tmp_path / f"out_{san}.bin")
    probe = subprocess.run([exe], capture_output=True, text=True, env=env)
    if probe.returncode != 2:   # (usage exit) e.g. a kernel that refuses the sanitizer's shadow mappings
        pytest.skip(f"the {san} runtime does not start here: rc {probe.returncode} {probe.stderr[-300:]}")
    r = subprocess.run([exe, path, out, "4"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    buf = np.fromfile(out, dtype=np.uint8)
    o = 0
    n_plans = 0
    for g, (pw, sw, n_blocks, scr) in zip(gs, want):
        npw = int(np.frombuffer(buf[o:o + 8], dtype=np.int64)[0]); o += 8
        got = np.frombuffer(buf[o:o + 4 * npw], dtype=np.int32); o += 4 * npw
        assert np.array_equal(got, pw), ("plan words differ", g[:5])
        nsw = int(np.frombuffer(buf[o:o + 8], dtype=np.int64)[0]); o += 8
        nb = int(np.frombuffer(buf[o:o + 4], dtype=np.int32)[0]); o += 4
        sc = int(np.frombuffer(buf[o:o + 8], dtype=np.int64)[0]); o += 8
        got = np.frombuffer(buf[o:o + 4 * nsw], dtype=np.int32); o += 4 * nsw
        assert nb == n_blocks and sc == scr and np.array_equal(got, sw), ("schedule differs", g[:5])
        n_plans += npw > 0
    assert o == len(buf) and n_plans >= 20     # (graphs whose tables do not fit a CU's LDS give no plan: still scheduled)
