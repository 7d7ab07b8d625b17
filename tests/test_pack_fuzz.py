"""CPU: the load-time walk of untrusted postings (csrc/mrk_pack.cpp: pack_term / validate_term -- what mrk_segment_create runs over
every doclist before any of it reaches a kernel) under AddressSanitizer + UBSan: damaged .spd bytes and damaged dictionary entries
in allocations of the exact size (tests/cpp/fuzz_pack.cpp).  Every walk ends in true or false-with-a-message; nothing is sized by
an unchecked count (a dictionary entry claiming 2^32 docs used to allocate tens of GB before the first byte was read)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.skipif(not shutil.which("g++"), reason="no g++")
def test_posting_walk_under_sanitizers(tmp_path):
    csrc = os.path.join(ROOT, "manticoresearch_amd", "csrc")
    exe = str(tmp_path / "fuzz_pack")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + csrc,
                           os.path.join(HERE, "cpp", "fuzz_pack.cpp"), os.path.join(csrc, "mrk_pack.cpp"), os.path.join(csrc, "mrk_writer.cpp"),
                           "-lpthread", "-o", exe])
    for seed in ("21", "22"):
        out = subprocess.run([exe, "8000", seed], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert out.returncode == 0, (out.stdout[-300:], out.stderr[-3000:])
        packed, declined = (int(x) for x in out.stdout.split()[1::2])
        assert packed + declined == 8000 * 6 * 2 and packed > 5000 and declined > 5000, out.stdout
