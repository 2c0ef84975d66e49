"""Golden vectors G10b: NeuS volume_render with `upsample_algo` = 'direct_use' / 'direct_more' (neus.py:242-269; YAML-reachable through
model.upsample_algo, neus.py:735), from the REAL reference (/root/reference) on CPU in the build container:

    python tests/golden/make_golden_neus_algos.py          -> tests/golden/neus_algos_golden.npz

Same stubs, scene and 8 x 8 camera as make_golden.py / make_golden_perturb.py.  Per algorithm: every `extras` key of
`volume_render(detailed_output=True)` at perturb=False, and - with torch.rand recorded (one [R, 64] draw per call, rend_util.py:272) -
rgb / depth / d_final at perturb=True.  (Kept apart from renderer_golden.npz so that file keeps regenerating bit for bit.)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden_perturb import RandRecorder  # noqa: E402


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from utils import io_util, rend_util
    from models.frameworks import get_model as ref_get_model
    import nerfart_amd  # noqa: F401
    from nerfart_amd import scene, frameworks
    torch.set_num_threads(8)
    out = {}
    cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", "neus_fangzhou_vangogh.yaml"))
    cfg.device_ids = [0]
    cfg.training.is_finetune = False
    torch.manual_seed(0)
    ref_model, _, rk_train, rk_test, ref_render = ref_get_model(cfg, [480, 270])
    torch.manual_seed(0)
    mine, _, _, _, _ = frameworks.get_model(scene.synthetic_config("NeuS"))
    sd = scene.perturb_state(mine.state_dict(), beta=None, seed=1)
    ref_model.load_state_dict(sd)
    out["A_state_sha256"] = np.array(mg.state_checksum(sd))
    H = W = 8
    c2w, K = scene.camera(H, W)
    ro, rd, _ = rend_util.get_rays(c2w[None], K[None], H, W)
    R = H * W
    out.update(A_c2w=c2w, A_K=K, A_H=np.array(H), A_W=np.array(W))
    for algo in ("direct_use", "direct_more"):
        rk = dict(rk_test)
        rk["upsample_algo"] = algo
        rk["perturb"] = False
        with torch.no_grad():
            rgb, depth, ex = ref_render(ro, rd, calc_normal=True, detailed_output=True, **rk)
        for k, v in ex.items():
            out[f"A_{algo}_{k}"] = v[0]
        rk["perturb"] = True
        with RandRecorder() as rr, torch.no_grad():
            torch.manual_seed(21)
            rgb, depth, ex = ref_render(ro, rd, calc_normal=True, detailed_output=True, **rk)
        assert len(rr.draws) == 1 and rr.draws[0].reshape(-1, 64).shape[0] == R, [tuple(x.shape) for x in rr.draws]
        out[f"A_{algo}_perturb_u"] = rr.draws[0].reshape(R, 64)
        for k in ("rgb", "depth_volume", "d_final", "mask_volume"):
            out[f"A_{algo}_perturb_{k}"] = ex[k][0]
        print(algo, "rgb mean", float(out[f"A_{algo}_rgb"].mean()), "keys", list(ex.keys()))
    np.savez_compressed(os.path.join(HERE, "neus_algos_golden.npz"), **mg.t2n(out))
    print("wrote neus_algos_golden.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
