import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

VOICE = ROOT / "tests" / "golden" / "voice" / "nitech_jp_atr503_m001.htsvoice"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_voice():
    from oracle import oracle as O

    return O.Voice(VOICE)
