"""The one-wave-per-SIMD K2 (csrc/mlp_k2_w32.hip, NERFART_K2=w32) stays correct: same SDF values as the oracle and as the default
8-wave kernel, ragged sizes, ray-source mode.  The switch is read once per process, so the kernel runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import scene_state
from oracle import nets
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="bf16x3")
blob, _ = model.packed()
sd, _ = scene_state("VolSDF", 0.01)
g = torch.Generator().manual_seed(3)
pts = torch.rand(3001, 3, generator=g) * 6 - 3
pts[:800] *= 0.3
worst = 0.0
for n in (1, 31, 32, 33, 127, 128, 129, 3001):
    p = pts[:n].contiguous()
    out = hip.sdf_fwd(blob, p.cuda(), 3.0, precision=1).cpu()
    ref = nets.volsdf_forward_surface(sd, p)[0]
    worst = max(worst, float((out - ref).abs().max()))
# rays + depths source, strided output (what the sampler launches)
o = torch.tensor([[0.0, 0.0, -2.5]]).expand(40, 3).contiguous()
d = torch.nn.functional.normalize(torch.randn(40, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).contiguous()
depth = torch.linspace(0.5, 4.0, 50)[None].expand(40, 50).contiguous()
out = hip.sdf_fwd_rays(blob, o.cuda(), d.cuda(), depth.cuda(), 3.0, precision=1).cpu()
ref = nets.volsdf_forward_surface(sd, (o[:, None] + d[:, None] * depth[..., None]).reshape(-1, 3))[0].reshape(40, 50)
worst = max(worst, float((out - ref).abs().max()))
print("WORST", worst)
'''


def _run(variant):
    env = dict(os.environ, NERFART_K2=variant)
    r = subprocess.run([sys.executable, "-c", CHILD % (REPO, os.path.join(REPO, "tests"))], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return float([l for l in r.stdout.splitlines() if l.startswith("WORST")][0].split()[1])


def test_w32_kernel_matches_oracle_like_the_default_kernel():
    w32, v1 = _run("w32"), _run("v1")
    print(f"  max |sdf - oracle|: w32 {w32:.2e}, default 8-wave kernel {v1:.2e}")
    assert w32 < 1e-4 and v1 < 1e-4            # the split-bf16 point-query tolerance of tests/test_gpu_bf16x3.py
