"""The reference's own mains compile, unmodified, against include/compat/ (the drop-in headers) -- fed to the compiler on
stdin so that no copy of them enters this repository.  Only where /root/reference exists (the build container)."""
import os
import subprocess

import pytest

from conftest import ROOT

REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference sources exist only in the build container")
@pytest.mark.parametrize("name", ["locator.cpp", "spaced_seed.cpp", "visual_align.cpp"])
def test_reference_main_compiles_against_compat_headers(name):
    src = open(os.path.join(REF_SRC, name), "rb").read()
    p = subprocess.run(["g++", "-fsyntax-only", "-w", "-x", "c++", "-", "-I", os.path.join(ROOT, "include", "compat"),
                        "-I", os.path.join(ROOT, "include")], input=src, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
