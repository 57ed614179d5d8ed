"""The wave-to-wave hand-off of the segmented march, stressed on its own (tools/ubench/xcd_handoff.hip): groups of words
pass from wave to wave through sc1 stores, a drained flag and sc1 loads, 24 times each, between CUs of different XCDs, with
cache lines shared between groups, five workgroups per CU and uneven arrivals -- and every word is checked.  The forms the
library uses (sc1 stores + sc1 loads, with or without an agent acquire after the poll) must show no stale word."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_segment_handoff_shows_no_stale_word(photon, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "xcd_handoff")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-Wno-unused-value", os.path.join(ROOT, "tools", "ubench", "xcd_handoff.hip"), "-o", exe],
                   check=True, timeout=300)
    out = subprocess.run([exe, "6000", "24"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120).stdout.decode()
    rows = re.findall(r"(aligned|shared ) lines, (.*?)\s*: (\d+) stale words of (\d+) checked, (\d+) waves gave up", out)
    assert len(rows) == 8, out
    for lines, form, stale, checked, gave_up in rows:
        assert int(gave_up) == 0 and int(checked) == 6000 * 24 * 64 * 8, (lines, form, out)
        if form.startswith("sc1 stores, sc1 loads"):             # what march_group does
            assert int(stale) == 0, (lines, form, out)
