import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu2: needs two MI355X (RCCL); always combined with `gpu`, skipped on a one-GPU box")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden_ops():
    return load_golden("ops.npz")


@pytest.fixture(scope="session")
def golden_modules():
    return load_golden("modules.npz")


@pytest.fixture(scope="session")
def golden_lbs():
    return load_golden("lbs.npz")


@pytest.fixture(scope="session")
def golden_smpl():
    return load_golden("smpl.npz")


@pytest.fixture(scope="session")
def golden_gcn():
    return load_golden("gcn.npz")


def sub_state_dict(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.fixture(params=["nvcc", "off"])
def contraction_mode(request):
    """Runs an index-parity test once per distance-contraction mode (include/g4d.h): the HIP library and the oracle are switched
    together, so each run compares the kernels with the MATCHING oracle.  Modules opt in with
    `pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("contraction_mode")]`."""
    from garment4d_amd import numerics
    from oracle import pointnet2_oracle as K
    prev_o = K.set_contraction(request.param)
    prev_l = numerics.set_distance_contraction(request.param)
    yield request.param
    numerics.set_distance_contraction(prev_l)
    K.set_contraction(prev_o)


@pytest.fixture
def request_finalizers(request):
    """A list of callables run (in reverse order) when the test ends, whatever its outcome."""
    fns = []
    yield fns
    for f in reversed(fns):
        f()


@pytest.fixture(scope="session")
def golden_refine():
    """tests/golden/refine.npz = outputs of the reference's own modules/mesh_encoder.py (make_golden_refine.py); the inputs are
    regenerated from the seed here and checked against the stored checksums."""
    from garment4d_amd import synthetic as syn
    g = load_golden("refine.npz")
    case = syn.refine_golden_case()
    assert np.array_equal(syn.refine_golden_checksum(case), g["checksum"]), "synthetic.refine_golden_case drifted from refine.npz"
    return g, case


@pytest.fixture
def tune():
    """tune(field=value, ...): run the REST of the test under a garment4d_amd.tuning.Tuning with these fields changed (native={"key": v} for the
    library's tuning table); may be called repeatedly (each call builds on the one before); everything is restored when the test ends.
    Replaces the monkeypatching of module-level switches of rounds 1-4: the kernel-selection state is one explicit object."""
    import contextlib
    from garment4d_amd import tuning
    stack = contextlib.ExitStack()

    def set_(**kw):
        stack.enter_context(tuning.use(tuning.current().replace(**kw)))
    yield set_
    stack.close()
