"""The C-ABI library loads on a CPU-only box, exports every symbol include/frx.h declares, and
fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def declared_symbols(header="frx.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(frx_[a-z_0-9]+)\s*\(", src)) - {"frx_batch_eval_fn"})


def test_every_declared_symbol_is_exported(frx):
    syms = declared_symbols()
    assert len(syms) >= 18
    L = C.CDLL(frx.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/frx.h but not exported by libfrx.so"
    assert sorted(frx.ABI_SYMBOLS) == syms
    assert frx.lib().frx_version() == 100
    # the boundary header reads as the reference's SE3GCOPTER / cuda_computer interface plus its neighbours: diagnostics live in frx_debug.h
    dbg = declared_symbols("frx_debug.h")
    assert sorted(frx.DEBUG_SYMBOLS) == dbg and not set(dbg) & set(syms)
    for s in dbg:
        assert hasattr(L, s), f"{s} declared in include/frx_debug.h but not exported by libfrx.so"
    assert not [s for s in syms if "debug" in s or "selftest" in s or "profile" in s]


def test_config_struct_layout_matches_header(frx):
    assert C.sizeof(frx.FrxConfig) == 3 * 8 + 2 * 4 + 8 * 8 + 4 * 8
    assert C.sizeof(frx.LbfgsParams) == 80


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_create_fails_loudly_without_device(frx, sc):
    assert frx.lib().frx_device_count() == 0
    with pytest.raises(frx.FrxError, match="no HIP device"):
        frx.Problem(sc.make_batch(0, 1, 8, 2), sc.ZHANGJIAJIE, qd_intervals=8)


def test_argument_validation(frx):
    h = C.c_void_p()
    z = np.zeros(1); zi = np.zeros(2, np.int32)
    cfg = frx.FrxConfig.from_params(__import__("fast_racing_amd").scenario.ZHANGJIAJIE)
    rc = frx.lib().frx_problem_create(C.byref(cfg), 0, 0, zi, z, z, zi, z, zi, z, C.byref(h))
    assert rc == -1 and b"B <= 0" in frx.lib().frx_last_error()


def test_product_never_references_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "fast-racing_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle/" not in txt.replace("the CPU oracle (oracle/Makefile)", "") and "liboracle" not in txt and "import oracle" not in txt, fn
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libfrx.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out


def test_cpp_mirror_header_compiles_and_links(frx, tmp_path):
    """include/se3gcopter_amd.hpp (the C++ mirror of SE3GCOPTER::{setup,optimize}) builds against libfrx.so; on a CPU-only box
    setup() must report the missing device instead of computing anything; on a GPU box the one-cell plan must succeed."""
    import subprocess
    exe = str(tmp_path / "integ_stub")
    pkg = os.path.join(ROOT, "fast-racing_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "integration_stub.cpp"), "-o", exe,
                    "-L" + pkg, "-lfrx", "-Wl,-rpath," + pkg], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
    assert ("no HIP device" in r.stdout) != has_gpu()


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_multi_create_fails_loudly_without_device(frx, sc):
    with pytest.raises(frx.FrxError, match="no HIP device"):
        frx.MultiProblem(sc.make_batch(0, 2, 8, 2), sc.ZHANGJIAJIE, qd_intervals=8)


def test_mailbox_threads_follow_the_cpu_share_of_the_plan(frx, monkeypatch):
    """the mailbox threads of a resident plan spin for its whole length, so their number follows the plan's share of the CPUs the
    process may use - cgroup quota / affinity, divided by the ranks of the node (LOCAL_WORLD_SIZE) and the shards of a frx_multi job - minus one for the
    rank's main thread, never more than one per sixteen clusters (at most four).  FRX_HOST_CPUS stands in for the quota here."""
    import ctypes as C
    L = frx.lib()

    def ask(clusters, extra=0):
        b, s, t = C.c_double(), C.c_int(), C.c_int()
        assert L.frx_debug_host_cpu_share(clusters, extra, C.byref(b), C.byref(s), C.byref(t)) == 0
        return b.value, s.value, t.value

    for var in ("FRX_LOCAL_RANKS", "LOCAL_WORLD_SIZE", "FRX_HOST_CPUS"):
        monkeypatch.delenv(var, raising=False)
    b, s, t = ask(32)
    assert b >= 1.0 and s == int(b) and 1 <= t <= 2
    monkeypatch.setenv("FRX_HOST_CPUS", "16")                       # the GPU boxes' quota
    assert ask(32) == (16.0, 16, 2) and ask(64)[2] == 4 and ask(1)[2] == 1      # one rank: one thread per sixteen clusters, at most four
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")                     # eight ranks of a node: two CPUs each - the caller alone serves its mailboxes
    assert ask(32) == (16.0, 2, 1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert ask(32) == (16.0, 4, 2) and ask(64)[2] == 3
    monkeypatch.setenv("FRX_LOCAL_RANKS", "2")                      # overrides the launcher's variable
    assert ask(64) == (16.0, 8, 4)
    monkeypatch.delenv("FRX_LOCAL_RANKS"); monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert ask(32, extra=7) == (16.0, 2, 1)                         # frx_multi: eight shards planning side by side in one process
    monkeypatch.setenv("FRX_HOST_CPUS", "1")
    assert ask(32)[1:] == (1, 1)
