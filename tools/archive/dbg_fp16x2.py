#!/usr/bin/env python
"""Debug aid for precision 4: blob (GPU packer vs CPU packer), kernel vs emulation vs oracle on a few points."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import emul_chain as em
from conftest import scene_state
from nerfart_amd import hip, packing, scene
from oracle import nets
DEV = "cuda:0"
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="fp16x2")
surf, rad = model.packed()
sd, _ = scene_state("VolSDF", 0.01)
surf_cpu = packing.surface_plan_bf16(term="fp16").pack(packing.surface_tensors(sd))
a, b = surf.cpu().numpy(), surf_cpu.numpy()
hdr = b[:512].view(np.int32)
body0, aux_off = 512, int(hdr[4])
ha = a[body0:aux_off].view(np.float16).astype(np.float32); hb = b[body0:aux_off].view(np.float16).astype(np.float32)
print("blob body: entries", ha.size, "differing bits", int((a[body0:aux_off].view(np.uint32) != b[body0:aux_off].view(np.uint32)).sum()), "max abs diff", float(np.abs(ha - hb).max()),
      "nonfinite gpu", int((~np.isfinite(ha)).sum()), "cpu", int((~np.isfinite(hb)).sum()), "max |w| gpu", float(np.abs(ha[np.isfinite(ha)]).max()))
print("aux max diff", float(np.abs(a[aux_off:] - b[aux_off:]).max()))
g = torch.Generator().manual_seed(23)
pts = (torch.rand(16, 3, generator=g) * 4 - 2)
x = pts.to(DEV)
for name, blob in (("gpu-packed", surf), ("cpu-packed", surf_cpu.to(DEV))):
    s4 = hip.sdf_fwd(blob, x, 3.0, precision=4).cpu().numpy()
    print(name, "K2 precision 4:", s4[:6])
m1, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
print("bf16x3 K2:", hip.sdf_fwd(m1.packed()[0], x, 3.0, precision=1).cpu().numpy()[:6])
em.TERM = "fp16"
print("emulation (cpu blob):", em.emul_sdf_only_bf16(b, pts.numpy(), 3.0)[:6])
em.TERM = "bf16"
s_ref = nets.volsdf_forward_surface(sd, pts)[0].numpy()
print("oracle:", s_ref[:6])
sdf, nab, h7 = hip.sdf_nabla_fwd(surf, x, 3.0, precision=4)
print("grad kernel sdf:", sdf.cpu().numpy()[:6], "nabla[0]:", nab[0].cpu().numpy(), "oracle nabla[0]:", nets.surface_forward_with_nablas(sd, pts)[1][0].numpy())
# ---- ray-mode entry points and the fused renderer
from nerfart_amd import rend_util
H, W = 8, 6
c2w, K = scene.camera(H, W)
o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
dn = hip.normalize_dirs(d[0].contiguous())
depth = torch.linspace(0.0, 6.0, 128, device=DEV)[None].expand(H * W, 128).contiguous()
s4 = hip.sdf_fwd_rays(surf, o[0].contiguous(), dn, depth, 3.0, precision=4)
s1 = hip.sdf_fwd_rays(m1.packed()[0], o[0].contiguous(), dn, depth, 3.0, precision=1)
print("sdf_fwd_rays p4 vs p1: max diff", float((s4 - s1).abs().max()), "finite", bool(torch.isfinite(s4).all()), s4[20, ::16].cpu().numpy(), s1[20, ::16].cpu().numpy())
for name, mdl in (("bf16x3", m1), ("fp16x2", model)):
    sb, rb = mdl.packed()
    out = hip.volsdf_render(sb, rb, 1, o[0].contiguous(), d[0].contiguous(), near=0.0, far=6.0, R_bg=3.0, alpha=100.0, beta=0.01, max_upsample_steps=6,
                            detailed=True, precision=mdl.precision_id)
    print(name, "rgb[20]", out["rgb"][20].cpu().numpy(), "depth", float(out["depth_volume"][20]), "acc", float(out["mask_volume"][20]), "iter", out["iter_usage"][:8].cpu().numpy(),
          "sdf range", float(out["implicit_surface"].min()), float(out["implicit_surface"].max()), "rad[20, 90]", out["radiance"][20, 90].cpu().numpy(),
          "nabla[20,90]", out["implicit_nablas"][20, 90].cpu().numpy())
