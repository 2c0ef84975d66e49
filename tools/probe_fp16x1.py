"""K2 at precisions 1 / 4 / 5 on 4 M random points (ms), sdf of 5 vs 4 vs 1, then a frame with the fp16x1 sampler."""
import sys, time, torch
sys.path.insert(0, '.')
from nerfart_amd import scene, hip, rend_util
dev = 'cuda:0'
m, rk, fn = scene.build_model('VolSDF', seed=0, beta=0.01, device=dev, precision='bf16x3')
g, v, b = m._surface_layers()
blob1 = hip.pack_surface_blob(1, 6, g, v, b); blob4 = hip.pack_surface_blob(4, 6, g, v, b)
torch.manual_seed(0)
x = (torch.rand(4 << 20, 3, device=dev) * 2 - 1) * 1.5
ref = hip.sdf_fwd(blob1, x, 3.0, precision=1)
m.set_sampler_precision('fp16x1', guard=0.05)
blob5 = m.packed_sampler()[0]
m.set_sampler_precision('fp16x1c', guard=0.005, late_round=3)
blob5c = m.packed_sampler()[0]
for name, blob, p in (('bf16x3', blob1, 1), ('fp16x2', blob4, 4), ('fp16x1 (nearest)', blob5, 5), ('fp16x1 (compensated)', blob5c, 5)):
    out = hip.sdf_fwd(blob, x, 3.0, precision=p)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = hip.sdf_fwd(blob, x, 3.0, precision=p); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    e = (out - ref).abs()
    print(f'{name}: {min(ts):.2f} ms / 4M points (median {sorted(ts)[2]:.2f}); vs bf16x3 sdf: max {float(e.max()):.2e} mean {float(e.mean()):.2e}', flush=True)
