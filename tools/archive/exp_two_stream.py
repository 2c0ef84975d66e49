#!/usr/bin/env python
"""Experiment: the two 65,536-ray chunks of a 480x270 frame rendered concurrently from two host threads on two
HIP streams (ctypes releases the GIL inside the C entry point, so one chunk's host read of its active-ray count does not stall the
other's launches) against the sequential chunk loop.  Prints ms per frame for both and whether the images are bit-identical."""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerfart_amd import scene, rend_util

dev = torch.device("cuda", 0)
model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="bf16x3")
H, W = 480, 270
c2w, K = scene.camera(H, W)
o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
kw = {k: v for k, v in rk.items() if k != "rayschunk"}
N = o.shape[1]


def seq():
    rgb, _, _ = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
    return rgb


def par(parts):
    bounds = [(N * i // parts, N * (i + 1) // parts) for i in range(parts)]
    out = [None] * parts
    streams = [torch.cuda.Stream() for _ in range(2)]
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)

    def work(t):
        with torch.cuda.stream(streams[t]):
            streams[t].wait_event(ev)
            for i in range(t, parts, 2):
                a, b = bounds[i]
                out[i] = render_fn(o[:, a:b].contiguous(), d[:, a:b].contiguous(), require_nablas=True, calc_normal=True, detailed_output=False, **kw)[0]
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for s in streams:
        main.wait_stream(s)
    return torch.cat(out, dim=1)


def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


res = {}
res["sequential_ms"], ref = timed(seq)
for parts in (2, 4, 8):
    ms, img = timed(lambda: par(parts))
    res[f"two_streams_{parts}_parts_ms"] = round(ms, 2)
    res[f"bit_identical_{parts}"] = bool(torch.equal(img, ref))
res["sequential_ms_again"], _ = timed(seq)
res = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}
print(json.dumps(res))
