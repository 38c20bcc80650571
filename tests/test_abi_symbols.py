"""The C-ABI library loads without a GPU and exports every symbol include/pba.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_functions(header="pba.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pba_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    from pacbioassembly_amd import _lib
    names = declared_functions()
    assert len(names) >= 35
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SYMBOLS) == names


def test_dist_header_symbols_exported(lib):
    """include/pba_dist.h (the exchange over RCCL for C / C++ hosts): libpba_dist.so loads and exports what it declares."""
    from pacbioassembly_amd import _lib
    names = declared_functions("pba_dist.h")
    assert len(names) >= 10 and all(n.startswith("pba_dist_") for n in names)
    raw = _lib.load_dist()
    assert not [n for n in names if not hasattr(raw, n)]
    assert sorted(_lib.DIST_SYMBOLS) == names
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    raw.pba_dist_shard(10, 1, 3, ctypes.byref(lo), ctypes.byref(hi))           # (host arithmetic: the same split as distributed.shard_range)
    assert (lo.value, hi.value) == (3, 6)


def test_no_device_is_an_error_not_a_fallback(lib):
    """Without a gfx950 GPU the device API refuses to start; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    st = lib.pba_ctx_create(0, ctypes.byref(h))
    assert st == -5 and not h.value          # PBA_E_NODEVICE


def test_product_does_not_link_the_oracle():
    """oracle/ is test infrastructure: nothing in the package may reference it."""
    pkg = os.path.join(ROOT, "pacbioassembly_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "pba_oracle" not in txt and "oraclelib" not in txt and "orc_" not in txt, f


def test_examples_build_against_the_c_abi_and_refuse_to_run_without_a_gpu(lib, tmp_path):
    """examples/*.cpp are plain C++ over include/pba.h: they compile and link with g++ (no HIP at the call site), and
    without a gfx950 device they stop with PBA_E_NODEVICE instead of computing anything on the CPU."""
    import subprocess
    import torch
    libdir = os.path.join(ROOT, "pacbioassembly_amd", "lib")
    for name in ("locator_gpu", "spaced_seed_gpu"):
        exe = str(tmp_path / name)
        subprocess.run(["g++", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
                        os.path.join(ROOT, "examples", name + ".cpp"), "-L", libdir, "-lpba", f"-Wl,-rpath,{libdir}"], check=True)
    # ... and the multi-GPU one over include/pba_dist.h (RCCL is libpba_dist.so's dependency, not the caller's)
    exe = str(tmp_path / "overlap_dist")
    subprocess.run(["g++", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "examples", "overlap_dist.cpp"),
                    "-L", libdir, "-lpba_dist", "-lpba", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    if torch.cuda.is_available():
        return
    r = subprocess.run([exe, "100", "1000"], capture_output=True)
    assert r.returncode != 0 and r.stdout == b"" and b"device" in r.stderr.lower()
    (tmp_path / "c.txt").write_text("ACGT" * 100 + "\n")
    (tmp_path / "s.txt").write_text("111*11*11*1*1111\n")
    (tmp_path / "r.bin").write_bytes(b"")
    r = subprocess.run([str(tmp_path / "locator_gpu"), str(tmp_path / "c.txt"), "111*11*11*1*1111"], input=b"ACGT\n", capture_output=True)
    assert r.returncode != 0 and r.stdout == b"" and b"device" in r.stderr.lower()
    r = subprocess.run([str(tmp_path / "spaced_seed_gpu"), "-f", str(tmp_path / "c.txt"), str(tmp_path / "r.bin"), str(tmp_path / "s.txt")],
                       capture_output=True)
    assert r.returncode != 0 and r.stdout == b""
