"""DESIGN.md's numbers table is generated from the committed profiles (tools/design_numbers.py): a table that has drifted from
the files under profiles/ fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_numbers_table_matches_the_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
