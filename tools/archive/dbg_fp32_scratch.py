import sys, torch
sys.path.insert(0, "/root/repo")
from nerfart_amd import scene, hip
model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device="cuda", precision="fp32")
surf, _ = model.packed()
g = torch.Generator().manual_seed(0)
M = 128 * 600
pts = (torch.rand(M, 3, generator=g) * 4 - 2).cuda()
s0, n0, h0 = hip.sdf_nabla_fwd(surf, pts, 3.0, precision=0)
s3, n3, h3 = hip.sdf_nabla_fwd(surf, pts, 3.0, precision=3)
bad = ((n0 - n3).abs().max(-1).values > 1e-4).cpu()
print("bad points", int(bad.sum()), "of", M, "; sdf max diff", float((s0 - s3).abs().max()), "h7", float((h0 - h3).abs().max()))
idx = bad.nonzero()[:, 0]
import collections
print("by wave (m//16 % 8):", sorted(collections.Counter(((idx // 16) % 8).tolist()).items()))
print("by column j (m % 16):", sorted(collections.Counter((idx % 16).tolist()).items()))
tiles = (idx // 128)
print("bad tiles:", len(set(tiles.tolist())), "of", M // 128, "first:", sorted(set(tiles.tolist()))[:20])
print("by tile % 256 (workgroup):", sorted(collections.Counter((tiles % 256).tolist()).items())[:12])
n0b, _, _ = hip.sdf_nabla_fwd(surf, pts, 3.0, precision=0)[1], None, None
print("repeatable:", bool(torch.equal(n0b, n0)))
