import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
MODELS = os.path.join(GOLDEN, 'models')

# The three cameras the reference's source holds for its shipped models (rendering.d:78, 81, 94), fov 30, up +z.
CAMERAS = {
    'cornell-box': ((-1.0, 26.5, 10.0), (0.0, 0.0, 9.0)),
    'teapot': ((25.0, 10.0, 20.0), (0.0, 0.0, 0.0)),
    'chest1': ((3.0, 3.0, 3.0), (0.0, 0.0, 0.0)),
}
UP = (0.0, 0.0, 1.0)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an MI355X (run with -m gpu on the GPU box)')


def model_path(name):
    return os.path.join(MODELS, name + '.thModel')


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope='session')
def flats():
    import thmodel_ref
    return {n: thmodel_ref.load_flat(model_path(n)) for n in CAMERAS}


def make_oracle(flat, name, W, H, build_tree=False, **cfg):
    import oracle_py
    o = oracle_py.Oracle(flat, build_tree=build_tree)
    o.configure(width=W, height=H, **cfg)
    o.set_camera(oracle_py.camera_lookat(30.0, H / W, CAMERAS[name][0], CAMERAS[name][1], UP))
    return o


def make_tracer(name, W, H, device=0, **cfg):
    import pathtracer_amd as P
    pt = P.PathTracer(W, H, device=device, **cfg)
    pt.Scene(model_path(name))
    pt.set_camera(P.Camera(30.0, H / W).setTransform(CAMERAS[name][0], CAMERAS[name][1], UP))
    return pt


@pytest.fixture(scope='session')
def require_gpu():
    if not gpu_available():
        pytest.skip('no GPU in this container (GPU tests run on the MI355X box)')


from pathtracer_amd.synthetic import soup  # noqa: E402,F401  (tests import it from here)
