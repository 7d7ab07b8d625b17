"""CPU: integration/mrk_replay.h -- the frame replay and the keyword-statistics order of the reference-side binding -- executed over
stub match / sorter types inside a restatement of MatchExtended's loop (tests/cpp/test_replay.cpp): the sorter's total ends at
total_found at end of stream, when the cutoff runs out on a frame's last row, and with an index weight; a repeated keyword is
reported once.  (mrk_adapter.h instantiates the same templates over the reference's types; tests/test_adapter_syntax.py.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_replay_and_word_stats(tmp_path):
    exe = str(tmp_path / "test_replay")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_replay.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "replay ok" in out.stdout, (out.stdout, out.stderr)
