"""CPU: the query planner (csrc/mrk_plan.cpp -- host code, no kernel in it) under AddressSanitizer + UBSan, fed flattened trees a
caller could hand to mrk_batch_submit: half of them well-formed (every operator, shared subtrees, filters, cutoffs), half hostile
(child indices out of range, cycles, unknown operators, keywords outside the dictionary, INT_MIN / INT_MAX arguments, NaN boosts,
impossible filter locators).  Every call must come back MRK_OK / MRK_E_UNSUPPORTED / MRK_E_INVAL, the passes and work items of an
accepted query must stay inside what the launch code indexes (tests/cpp/fuzz_plan.cpp), and neither sanitizer may fire."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc builds the host-only objects")
def test_planner_under_sanitizers(tmp_path):
    flags = ["-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    objs = []
    for src in (os.path.join(ROOT, "manticoresearch_amd", "csrc", "mrk_plan.cpp"), os.path.join(HERE, "cpp", "fuzz_plan.cpp")):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call([HIPCC] + flags + ["-c", src, "-o", obj])
        objs.append(obj)
    exe = str(tmp_path / "fuzz_plan")
    subprocess.check_call([HIPCC, "-fsanitize=address,undefined"] + objs + ["-o", exe])
    for seed in ("11", "12"):
        out = subprocess.run([exe, "250000", seed], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
        ok, uns, inval = (int(x) for x in out.stdout.split()[1::2])
        assert ok + uns + inval == 250000 and ok > 10000 and uns > 10000 and inval > 10000, out.stdout
