import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "multi-gpu-path-tracer_amd"
BOX_CAMERA = ((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0), 45.0, 45.0)     # SURVEY.md 8d convention
DUCK_CAMERA = ((0.0, 0.0, 0.5), (0.0, 0.0, -0.5), 45.0, 45.0)           # main.cu:40


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pt():
    """The product package (loads libptamd.so; building it first if the tree is fresh)."""
    mod = importlib.import_module(PKG)
    if not os.path.exists(mod.capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    mod.capi.lib()
    return mod


@pytest.fixture(scope="session")
def ob():
    """The oracle binding (test infrastructure)."""
    from oracle import binding
    if not os.path.exists(binding.LIB_PATH):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def box_scene(pt):
    return pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention()


@pytest.fixture(scope="session")
def duck_scene(pt):
    return pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_duck.glb"))


@pytest.fixture(scope="session")
def box_oracle(ob, box_scene):
    return ob.OracleScene(box_scene.triangles, box_scene.materials, box_scene.textures)


@pytest.fixture(scope="session")
def duck_oracle(ob, duck_scene):
    return ob.OracleScene(duck_scene.triangles, duck_scene.materials, duck_scene.textures)


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; none visible")
    return 0
