import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) — run via gpurun")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build libkta_hip.so / the oracle once if they are missing (hipcc cross-compiles without a GPU)."""
    from kafka_topic_analyzer_amd import build
    build.build_all()
