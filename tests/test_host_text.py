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


def _build_io_seams(tmp_path, sanitizer):
    exe = str(tmp_path / ("io_seams_" + sanitizer.replace(",", "_")))
    csrc = os.path.join(ROOT, "quade_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "native", "io_seams_test.cpp"),
           os.path.join(csrc, "quade_io.cpp"), os.path.join(csrc, "fastq_pack.cpp"), os.path.join(csrc, "quade_pgz.cpp"), "-o", exe, "-lz", "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_io_seams_of_the_pipeline_under_asan_and_ubsan(tmp_path):
    """quade_io_internal.h (raw text reader on every input flavour, host inflate of BGZF members, the sink's files, the pool) -- what
    quade_pipe.cpp builds on -- with AddressSanitizer + UBSan, no GPU."""
    exe = _build_io_seams(tmp_path, "address,undefined")
    d = tmp_path / "w"
    d.mkdir()
    # (leak check off: the library keeps one libdeflate compressor / decompressor per thread and level for the life of the process)
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-3000:])


def test_io_seams_of_the_pipeline_under_tsan(tmp_path):
    exe = _build_io_seams(tmp_path, "thread")
    d = tmp_path / "w"
    d.mkdir()
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-3000:])


def test_inflate3_lane_decoder_against_zlib(tmp_path):
    """tests/native/inflate3_lane_test.cpp: the third inflater's per-lane DEFLATE decoder (quade_amd/csrc/inflate3_lane.h compiles for the
    host too) against zlib -- every block type, levels and strategies, the input ring's "only landed bytes are taken" rule, units cut at
    block boundaries, a stop position that is no block boundary, damaged and truncated streams."""
    exe = str(tmp_path / "inflate3_lane_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "native", "inflate3_lane_test.cpp"), "-lz"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-2000:]


def test_the_gzip_probes_text_filter_names_exactly_the_bytes_text_does_not_hold():
    """quade_inflate3.hip: gz_text_lengths_only refuses a block header whose literal code covers a byte that fastq text cannot hold.
    The 256-bit set in the source must be everything but tab, newline, carriage return and 32 .. 126 -- a wrong bit would refuse
    real blocks (slow: every such stream pays the second, plain probe) or let false headers through."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quade_amd", "csrc", "quade_inflate3.hip")).read()
    m = re.search(r"non_text\[4\]\s*=\s*\{([^}]*)\}", src)
    assert m, "the filter's table moved"
    words = []
    for tok in m.group(1).split(","):
        tok = tok.strip()
        words.append(0xFFFFFFFFFFFFFFFF if tok == "~0ull" else int(tok.rstrip("ul"), 16))
    assert len(words) == 4
    in_set = {b for b in range(256) if (words[b >> 6] >> (b & 63)) & 1}
    text = {9, 10, 13} | set(range(32, 127))
    assert in_set == set(range(256)) - text
    # every byte of a fastq record is text in this sense
    assert all(b in text for b in b"@SIM:1:FC:12:34 1:N:0:ACGT\nACGTN\n+\nFF:,#IJ~!\r\n")
