#!/usr/bin/env python
"""How much of k_sdf_grad_bf16's softplus' scratch is saturated?  The reverse-mode grad(SDF) kernel parks
sigma'(z_l) = sigmoid(100 z_l) of layers 0..6 as unorm16 (3.5 KB per point, the 10.9 GB round trip of a 1.555 M-point launch).  With
softplus beta = 100 most pre-activations are far from 0: this measures, on the benchmark scene at the points the kernel really sees
(the 192 final samples of every ray of the 480 x 270 frame, strided), the share of stored values that are exactly 0, exactly 65535, and
in between - per layer, and per 8-value group (the granularity a class + payload layout would work at: one group = the 8 features a
lane holds of one unit).

    python tools/sigma_prime_hist.py > profiles/rNN_sigma_prime_hist.json
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    from nerfart_amd import scene, rend_util, autodiff, packing
    dev = "cuda:0"
    H, W = 480, 270
    out = {"what": "unorm16 softplus'(z_l) = round(65535 sigmoid(100 z_l)) at the 192 final samples of every 16th ray of the 480 x 270 benchmark frame", "scenes": {}}
    perm = torch.tensor([packing.unit_feature_hidden(u, g, e) for u in range(8) for g in range(4) for e in range(8)], device=dev)
    for beta in (0.01, 0.002):
        model, rk, fn = scene.build_model("VolSDF", seed=0, beta=beta, device=dev, precision="bf16x3")
        kw = {k: v for k, v in rk.items() if k != "rayschunk"}
        c2w, K = scene.camera(H, W)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        sel = torch.arange(0, H * W, 16, device=dev)
        _, _, ex = fn(o[:, sel], d[:, sel], require_nablas=True, calc_normal=True, detailed_output=True, **kw)
        dv = ex["d_vals"][0]
        dn = torch.nn.functional.normalize(d[0, sel], dim=-1)
        x = (o[0, sel][:, None, :] + dn[:, None, :] * dv[:, :, None]).reshape(-1, 3)
        surf = model.implicit_surface
        rec = {"points": int(x.shape[0]), "layers": {}}
        tot = np.zeros(3)
        grp = np.zeros(4)            # 8-value groups: all zero, all one, all saturated (mixed 0 / 1), with >= 1 interior value
        with torch.no_grad():
            e = autodiff.embed(x, surf.embed_multires)
            h = e
            for l in range(surf.D):
                if l in surf.skips:
                    h = torch.cat([h, e], dim=-1) / np.sqrt(2)
                z = autodiff.wn_linear(surf.surface_fc_layers[l], h)
                q = torch.round(65535.0 * torch.sigmoid(100.0 * z))
                if l < 7:                                             # layers 0..6 are parked; layer 7's derivative is consumed in registers
                    n = q.numel()
                    c = np.array([float((q == 0).sum()), float((q == 65535).sum()), 0.0])
                    c[2] = n - c[0] - c[1]
                    if q.shape[1] < 256:                              # layer 3 is 217 wide: its unit 7 slots are constant zeros in the kernel
                        q = torch.cat([q, torch.zeros(q.shape[0], 256 - q.shape[1], device=dev)], dim=1)
                    g8 = q[:, perm].reshape(q.shape[0], 32, 8)        # unit order: 8 consecutive slots = one lane's 16 bytes
                    z0, z1 = (g8 == 0), (g8 == 65535)
                    allz, allo = z0.all(-1), z1.all(-1)
                    sat = (z0 | z1).all(-1)
                    gc = np.array([float(allz.sum()), float(allo.sum()), float((sat & ~allz & ~allo).sum()), float((~sat).sum())])
                    rec["layers"][str(l)] = {"exactly_0": round(c[0] / n, 4), "exactly_1": round(c[1] / n, 4), "in_between": round(c[2] / n, 4),
                                             "groups_of_8": {"all_0": round(gc[0] / gc.sum(), 4), "all_1": round(gc[1] / gc.sum(), 4),
                                                             "saturated_mixed": round(gc[2] / gc.sum(), 4), "has_interior": round(gc[3] / gc.sum(), 4)}}
                    tot += c
                    grp += gc
                h = autodiff.softplus100(z)
        rec["all_layers"] = {"exactly_0": round(tot[0] / tot.sum(), 4), "exactly_1": round(tot[1] / tot.sum(), 4), "in_between": round(tot[2] / tot.sum(), 4),
                             "saturated": round((tot[0] + tot[1]) / tot.sum(), 4),
                             "groups_of_8": {"all_0": round(grp[0] / grp.sum(), 4), "all_1": round(grp[1] / grp.sum(), 4),
                                             "saturated_mixed": round(grp[2] / grp.sum(), 4), "has_interior": round(grp[3] / grp.sum(), 4)}}
        # bytes per point of a "2-bit class per value + 16-bit payload for the interior values" layout against the 3,584 B stored today
        interior = tot[2] / tot.sum()
        rec["bytes_per_point"] = {"today_unorm16": 7 * 256 * 2, "class2bit_plus_payload_ideal": round(7 * 256 * (0.25 + 2.0 * interior), 1),
                                  "group_class_plus_16B_for_interior_groups": round(7 * 32 * (0.25 + 16.0 * grp[3] / grp.sum()), 1)}
        out["scenes"][f"beta_{beta}"] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
