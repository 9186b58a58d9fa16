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


# Output of the native headless driver (csrc/headless_main.cpp), run once as a child process at session start: a
# process that has already touched the GPU must not start other programs on this pool, so it cannot be started from
# inside a GPU test.  None where there is no GPU or the driver is not built.
HEADLESS = None
HEADLESS_ARGS = dict(model='cornell-box', W=320, H=240, passes=4, seed=12)


def run_headless(out_dir):
    """ptgpu_headless on the cornell box at the reference's default frame size (rendering.d:19-20), 1 sample per
    pass as main.d:273-287 does; returns its exit code, stdout and the paths of the PPM and project file it wrote."""
    import subprocess
    exe = os.path.join(ROOT, 'pathtracer_amd', 'ptgpu_headless')
    a = HEADLESS_ARGS
    ppm, proj = os.path.join(out_dir, 'headless.ppm'), os.path.join(out_dir, 'sponza.project')
    cmd = [exe, os.path.join(MODELS, a['model'] + '.thModel'), '--size', str(a['W']), str(a['H']), '--passes', str(a['passes']),
           '--seed', str(a['seed']), '--out', ppm, '--project', proj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True)
    return dict(rc=r.returncode, stdout=r.stdout, ppm=ppm, project=proj)


def pytest_sessionstart(session):
    global HEADLESS
    exe = os.path.join(ROOT, 'pathtracer_amd', 'ptgpu_headless')
    if os.path.exists('/dev/kfd') and os.path.exists(exe):
        import tempfile
        try:
            HEADLESS = run_headless(tempfile.mkdtemp(prefix='ptgpu_headless_'))
        except Exception as e:                                    # reported by the test that looks at it
            HEADLESS = dict(rc=-1, stdout='could not run: %r' % (e,), ppm=None, project=None)


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
