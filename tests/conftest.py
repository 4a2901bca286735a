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


def tohoku_voices():
    """The two voice files of the reference's only multi-voice golden (`bonsai_multi`, src/lib.rs:77-91):
    models/tohoku-f01/tohoku-f01-{neutral,happy}.htsvoice.  The reference tree carries them as an un-fetched git
    submodule (.gitmodules:1-3) and nothing here fetches anything: point JB_TOHOKU_DIR at a directory holding the
    two files (or drop them into tests/golden/voice/tohoku-f01/) and the skipped tests run."""
    cands = [Path(os.environ["JB_TOHOKU_DIR"])] if os.environ.get("JB_TOHOKU_DIR") else []
    cands.append(ROOT / "tests" / "golden" / "voice" / "tohoku-f01")
    for d in cands:
        paths = [d / "tohoku-f01-neutral.htsvoice", d / "tohoku-f01-happy.htsvoice"]
        if all(p.is_file() for p in paths):
            return paths
    pytest.skip("tohoku-f01-neutral/happy.htsvoice not found (the reference's models/tohoku-f01 submodule is empty): "
                "set JB_TOHOKU_DIR to the directory that holds them to pin the two-voice blend")


# Engine::load([neutral, happy]) + the weights of src/lib.rs:80-84 (GV weights stay at InterporationWeight::new's
# equal split, src/model/interporation_weight.rs:48-60)
BONSAI_MULTI_WEIGHTS = {"duration": [0.7, 0.3], "parameter": [[0.7, 0.3], [0.7, 0.3], [1.0, 0.0]]}
# src/model/mod.rs:395-428 `multiple_models`: neutral + happy, set_duration([0.7, 0.3]), set_parameter(1, [0.7, 0.3]),
# label SAMPLE_SENTENCE_1[2]: Models::duration() and Models::stream(1)[0], compared with assert_eq! (exact)
MULTIPLE_MODELS_WEIGHTS = {"duration": [0.7, 0.3], "parameter": [[0.5, 0.5], [0.7, 0.3], [0.5, 0.5]]}
MULTIPLE_MODELS_DURATION = [(3.345043873786926, 6.943870377540589), (9.866290760040282, 59.23959312438964),
                            (5.616884994506836, 16.154539680480955), (1.7678393721580503, 0.9487730085849762),
                            (1.3566675186157227, 1.2509666562080382)]
MULTIPLE_MODELS_LF0_STATE0 = ([(5.354794883728027, 0.00590993594378233), (-0.004957371624186635, 0.00017984867736231536),
                               (0.010301648452877997, 0.00044686400215141473)], 0.9955164790153503)
BONSAI_MULTI_GOLDEN = {"len": 74880, 2000: 2.3158134981607754e-5, 30000: 6459.375032316974}  # src/lib.rs:88-90
