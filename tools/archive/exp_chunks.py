import sys, time, torch
sys.path.insert(0, "/root/repo")
from nerfart_amd import scene, rend_util
dev = "cuda"
model, rk, fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="mixed")
H, W = 480, 270
kw = {k: v for k, v in rk.items() if k != "rayschunk"}
angles = scene.spiral(90)
views = []
for s in range(5):
    c2w, K = scene.camera(H, W, angle=angles[(7 * s + 3) % 90])
    views.append(rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)[:2])
ref = None
for rc, k3 in ((65536, 8192), (131072, 8192), (131072, 16384), (131072, 32768), (65536, 16384), (65536, 8192)):
    fn(*views[0], require_nablas=True, calc_normal=True, detailed_output=False, rayschunk=rc, honor_rayschunk=True, k3_rays_chunk=k3, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for o, d in views[1:]:
        rgb, _, _ = fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, rayschunk=rc, honor_rayschunk=True, k3_rays_chunk=k3, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    if ref is None: ref = rgb.clone()
    print(rc, k3, round(dt * 1e3, 2), "ms", bool(torch.equal(rgb, ref)))
