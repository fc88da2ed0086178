"""CPU: the record equals the measurements -- tools/check_docs.py (headline figures of DESIGN.md / README.md against the newest committed bench line, cited
profile files exist)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_and_readme_agree_with_the_newest_bench_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_docs.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
