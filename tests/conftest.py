import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
URDF = os.path.join(ROOT, "numbotics_amd", "models", "kinova_cyl.urdf")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def g12():
    return np.load(os.path.join(GOLDEN, "g1_g2_kernels.npz"))


@pytest.fixture(scope="session")
def g3():
    return np.load(os.path.join(GOLDEN, "g3_kinova.npz"))


@pytest.fixture(scope="session")
def g5():
    return np.load(os.path.join(GOLDEN, "g5_connector.npz"))


@pytest.fixture()
def fresh_world():
    from numbotics_amd.physics import World
    from numbotics_amd.physics.world import _reset_worlds
    _reset_worlds()
    w = World()
    yield w
    _reset_worlds()


@pytest.fixture()
def kinova(fresh_world):
    """(arm, chain, [obstacles]) : Kinova-like fixture + the README cube (README.md:96)."""
    from numbotics_amd.scenes import build_scene
    return build_scene("c2")
