import json
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = ["cornell_box", "cornell_box_small_lights", "cornell_box_with_volume"]   # the BASELINE configs
# beyond BASELINE: sphere lights + metal, dielectric, a room-filling volume
EXTRA_SCENES = ["light_test", "three_orbs", "cornell_box_with_volume2", "cornell_box_nested_fog"]   # the last: a medium inside a medium (volume.h:10)
# checker / perlin textures, textured emitter and textured World::background
TEXTURE_SCENES = ["cornell_box_image_light", "textured_room", "image_room"]
ALL_SCENES = SCENES + EXTRA_SCENES + TEXTURE_SCENES
ORACLE_SCENES = ALL_SCENES


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pt_create builds the traversal kernels once more per scene (hiprtc, 1.5 - 3 s, on a thread the context joins when it
    # is destroyed).  The suite creates hundreds of short-lived contexts: the per-scene build is off here and switched on by
    # the tests that hold it to the same bits (tests/test_gpu_spec.py, tests/test_spec_build.py); bench.py always uses it.
    os.environ.setdefault("PATHTRACE_HIP_SPEC", "off")


def scene_path(name):
    return os.path.join(ROOT, "scenes", name + ".json")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on first use."""
    from oracle import pt_oracle

    pt_oracle.build()
    return pt_oracle
