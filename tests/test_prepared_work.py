"""Work prepared without a device (tools/next_round/) must stay usable until it has been tried."""

import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prepared_patch_still_applies():
    """tools/next_round/midpoint_table.patch (tools/next_round/README.md) applies to the tree."""
    patch = os.path.join(ROOT, 'tools', 'next_round', 'midpoint_table.patch')
    if not os.path.exists(patch) or not os.path.isdir(os.path.join(ROOT, '.git')):
        pytest.skip('no prepared patch / not a git checkout')
    out = subprocess.run(['git', 'apply', '--check', patch], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
