"""CPU, build container only: integration/mrk_adapter.h -- the ISphRanker adapter, FlattenXQ and the eligibility test a
maintainer adds to the reference -- must compile against the reference's own headers (syntax only; nothing is copied or
linked).  Skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree only exists in the build container")
def test_adapter_compiles_against_the_reference_headers():
    r = subprocess.run(["bash", os.path.join(ROOT, "integration", "check_adapter.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "compiles against" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
