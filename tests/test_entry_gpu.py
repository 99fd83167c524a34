"""The driver's round-end entry point must stay green: run __graft_entry__.smoke() as a GPU test."""
import pytest

pytestmark = pytest.mark.gpu


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()
