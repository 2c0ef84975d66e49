import hashlib
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden", "renderer_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # EXPERIMENT switch of the test suite only (VERDICT r05 next 3 i; profiles/r08_radiance_fp16x2_battery.log): NERFART_TEST_RADIANCE=fp16x2 puts every
    # split-bf16 / mixed VolSDF model the tests build onto the 2-MFMA radiance kernels (model.set_radiance_precision), to see which assertion of the
    # battery the cheaper radiance arithmetic breaks.  Never set in a judged run; the product has no such switch.
    exp = os.environ.get("NERFART_TEST_RADIANCE")
    if exp:
        from nerfart_amd import nets
        orig = nets._PackedModel.set_precision

        def patched(self, precision):
            orig(self, precision)
            if hasattr(self, "ln_beta") and self.precision == "bf16x3":
                self.set_radiance_precision(exp)
            return self
        nets._PackedModel.set_precision = patched


def state_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


@pytest.fixture(scope="session")
def golden():
    z = np.load(GOLDEN, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def perturb_golden():
    """perturb=True vectors (tests/golden/make_golden_perturb.py)."""
    z = np.load(os.path.join(os.path.dirname(GOLDEN), "perturb_golden.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


_states = {}


def scene_state(framework="VolSDF", beta=0.01):
    """Reference-format state dict of the synthetic scene (regenerated from seeds, CPU tensors)."""
    key = (framework, beta)
    if key not in _states:
        from nerfart_amd import scene, frameworks
        torch.manual_seed(0)
        model, _, _, rk_test, _ = frameworks.get_model(scene.synthetic_config(framework))
        _states[key] = (scene.perturb_state(model.state_dict(), beta=beta, seed=1), dict(rk_test))
    return _states[key]


@pytest.fixture(scope="session")
def volsdf_state(golden):
    sd, rk = scene_state("VolSDF", 0.01)
    assert state_checksum(sd) == str(golden["G3_state_sha256"]), \
        "regenerated synthetic weights differ from the ones the golden vectors were captured with (RNG drift)"
    return sd, rk


@pytest.fixture(scope="session")
def neus_state(golden):
    sd, rk = scene_state("NeuS", None)
    assert state_checksum(sd) == str(golden["G10_state_sha256"])
    return sd, rk


def tt(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="session")
def neus_algos_golden():
    """NeuS upsample_algo = direct_use / direct_more vectors (tests/golden/make_golden_neus_algos.py)."""
    z = np.load(os.path.join(os.path.dirname(GOLDEN), "neus_algos_golden.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}
