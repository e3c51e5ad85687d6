"""No GPU: the arithmetic and the per-record rules the device text stages share with the host (quade_amd/csrc/text_rules.h, crc_lds.h), built with g++
and checked against zlib and the rules they restate."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shared_text_rules_and_crc_arithmetic(tmp_path):
    exe = str(tmp_path / "text_rules_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "native", "text_rules_test.cpp"), "-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:]
