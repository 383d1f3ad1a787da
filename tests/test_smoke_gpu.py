"""The driver's smoke entry point must stay green: run it as part of the GPU suite."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.gpu
def test_graft_entry_smoke():
    import __graft_entry__ as entry
    entry.smoke()
