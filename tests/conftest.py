import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEFAULT_CFG = {
    "fg_exist_thr": 0.001,
    "bg_mask": {"thr": 25, "fusion_weight": 0.1},
    "objectremoval": {"score_map_center": {"landscape": [0.5, 0.5], "portrait": [0.6, 0.5]},
                      "saliency_thr": 0.00001, "consensus_thr": 0.5},
    "trimap": {"input_long_side": 960, "kernelsize": 3, "iters": 5, "color_winsize": [10, 100, 180]},
    "vmatting": {"model_path": "synthetic", "input_long_side": 960},
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cfg():
    import copy
    return copy.deepcopy(DEFAULT_CFG)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
