"""Host code under AddressSanitizer + UBSan (CPU build only -- GPU ASan is not available on this pool).

tests/cpp/san_host.cpp is compiled together with the product's host translation units (pba_codec.cpp, pba_synth.cpp)
and the CPU oracle (oracle/pba_oracle.c) with -fsanitize=address,undefined -fno-sanitize-recover=all and run on the
codec / aligner goldens (exported below to a flat text file) and on seeded random inputs in exact-size heap buffers."""
import os
import shutil
import subprocess

import pytest

from conftest import HERE, ROOT, gold_json

BUILD = os.path.join(HERE, "cpp", "_build")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def hexs(b: bytes) -> str:
    return b.hex() if b else "-"


def export_vectors(path):
    g = gold_json("codec.json")
    with open(path, "w") as f:
        for w, code in g["encode"]:
            f.write(f"ENC {hexs(w.encode('latin1'))} {code}\n")
        for c, v in g["c2i"]:
            f.write(f"C2I {c} {v}\n")
        f.write(f"SEEDTEXT {hexs(g['seed_at_text'].encode())}\n")
        for pos, want in g["seed_at"]:
            f.write(f"SEEDAT {pos} {want}\n")
        for pat, m in g["masks"]:
            f.write(f"MASK {hexs(pat.encode())} {m}\n")
        for s, hexrec, back in g["text2bin"]:
            f.write(f"T2B {hexs(s.encode())} {hexrec} {hexs(back.encode())}\n")
        for c in gold_json("align_kat.json"):
            e = c["exp"]
            f.write("ALN %s %s %.17g %d %d %d %d %d %d %d %d %d %d %d\n" % (
                hexs(c["a"].encode("latin1")), hexs(c["b"].encode("latin1")), c["R"], c["a_fwd"], c["b_fwd"], e["rc"],
                e.get("cost", 0), e.get("matlen_a", 0), e.get("matlen_b", 0), e["len_a"], e["len_b"], e["max_dst"],
                e.get("nedit", 0), e.get("first_op", 0)))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="needs gcc / g++")
def test_host_code_under_asan_ubsan():
    os.makedirs(BUILD, exist_ok=True)
    csrc = os.path.join(ROOT, "pacbioassembly_amd", "csrc")
    inc = ["-I", os.path.join(ROOT, "include"), "-I", csrc, "-I", os.path.join(ROOT, "oracle")]
    orc_o = os.path.join(BUILD, "pba_oracle_san.o")
    exe = os.path.join(BUILD, "san_host")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", *SAN, "-c", os.path.join(ROOT, "oracle", "pba_oracle.c"), "-o", orc_o],
                   check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", *SAN, *inc, os.path.join(HERE, "cpp", "san_host.cpp"),
                    os.path.join(csrc, "pba_codec.cpp"), os.path.join(csrc, "pba_synth.cpp"), orc_o, "-o", exe, "-lpthread"],
                   check=True)
    vec = os.path.join(BUILD, "san_vectors.txt")
    export_vectors(vec)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([exe, vec], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "san_host ok" in out, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
