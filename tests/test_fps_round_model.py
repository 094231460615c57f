"""The round rule of csrc/fps.hip, restated on the CPU (tools/fps_round_model.py): candidate hierarchy (thread -> wave ->
workgroup), bound B, exact chain on the candidate set.  The script asserts that its sample sequence equals the plain
sequential arg-max loop; here it runs on a small cloud for both rules and for the unsorted candidate set the kernel uses."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [["--rule", "prefix", "--mg", "32"],
                                   ["--rule", "greedy", "--eligible", "8"],
                                   ["--rule", "greedy", "--eligible", "16", "--m", "16", "--per-thread", "2"],
                                   ["--rule", "greedy", "--eligible", "32", "--m", "32", "--mw", "8", "--per-thread", "2",
                                    "--order", "hilbert", "--fold-stats"]])
def test_round_rules_reproduce_sequential_fps(extra):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fps_round_model.py"), "--n", "20000", "--k", "1500", "--g", "4",
           "--check", "1500"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "'rounds':" in res.stdout
    if "--fold-stats" in extra:  # the fold's critical path with and without the second test per pair of rows
        assert "'with_row_pair_tests':" in res.stdout and "'order': 'hilbert'" in res.stdout
