#!/usr/bin/env python
"""Fine-tune step (BASELINE configs[2] / SURVEY cfg3) timing on one MI355X: 480x270 rays, VolSDF dims, random-weight
CLIP ViT-B/32, perturb=False.  pass 1 = HIP renderer (staged entries, per-point state kept for pass 2; --no-keep: the
fused renderer, only the depths kept); style loss = CLIP directional + contrastive + PatchNCE (text features cached);
pass 2 = hand-written backward kernels + weight-gradient GEMMs, --patches-per-launch reference patches of 1200 rays per
launch group; Adam step.
Prints one JSON line (NOT the driver's bench contract - that is bench.py)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=270)
    ap.add_argument("--pass2-rays", type=int, default=1200)
    ap.add_argument("--patches-per-launch", type=int, default=4)
    ap.add_argument("--no-keep", action="store_true")
    ap.add_argument("--no-vgg", action="store_true", help="leave the VGG16 perceptual term (random weights) out of the style loss")
    args = ap.parse_args()
    from nerfart_amd import scene, rend_util, criteria, clip_vit, vgg
    from nerfart_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="bf16x3")
    H, W = args.H, args.W
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
    feats = criteria.ClipFeatures(model=clip_vit.build_clip(dev, seed=0), device=dev, synthetic=True)
    style = criteria.StyleLoss(feats, (H, W), neg_texts=[f"negative prompt {i}" for i in range(16)],
                               perceptual=None if args.no_vgg else vgg.VGGPerceptualLoss().to(dev))
    with torch.no_grad():
        target, _, _ = render_fn(o, d, detailed_output=False, require_nablas=True, calc_normal=True, **{k: v for k, v in rk.items() if k != "rayschunk"})
    # the "photo" the render is compared with: the render itself, low-pass perturbed (pred == gt would make the
    # directional loss 0/0, as in the reference)
    g = torch.Generator(device="cpu").manual_seed(0)
    noise = torch.nn.functional.interpolate(torch.randn(1, 3, H // 8, W // 8, generator=g), size=(H, W), mode="bicubic", align_corners=False)
    target = (target.reshape(1, H, W, 3) + 0.1 * noise.permute(0, 2, 3, 1).to(dev)).clamp(0, 1).reshape(1, -1, 3)
    tr = Trainer(model, pass2_rays=args.pass2_rays, patches_per_launch=args.patches_per_launch)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    times = []
    for it in range(args.steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if args.no_keep:
            rgb, depths_all = tr.render_image(render_fn, o, d, want_depths=True, **rk)
            kept = None
        else:
            rgb, depths_all = tr.render_keep(o, d, **rk), None
            kept, tr._kept = tr._kept, None
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rgb = rgb.detach().reshape(1, -1, 3).requires_grad_(True)
        to_img = lambda t: t.reshape(1, H, W, 3).permute(0, 3, 1, 2)
        loss = style(to_img(rgb), to_img(target.reshape(1, -1, 3)))
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.zero_grad()
        eik = tr.backward_patches(o, d, rgb.grad.detach()[0], depths_all=depths_all, kept=kept, **rk)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        if it > 0:
            times.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    m = [sum(x[i] for x in times) / len(times) for i in range(4)]
    print(json.dumps({"workload": f"fine-tune step {H}x{W}, VolSDF 128+64 spp, CLIP ViT-B/32 + VGG16 random weights", "steps": args.steps, "pass1_state_kept": not args.no_keep, "patches_per_launch": args.patches_per_launch, "vgg_perceptual_term": not args.no_vgg,
                      "s_per_step": round(sum(m), 3), "pass1_render_s": round(m[0], 3), "style_losses_fwd_bwd_s": round(m[1], 3),
                      "pass2_sampler_autograd_s": round(m[2], 3), "adam_s": round(m[3], 4), "loss": float(loss), "eikonal": eik,
                      "rays_per_s": round(H * W / sum(m), 1), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))


if __name__ == "__main__":
    main()
