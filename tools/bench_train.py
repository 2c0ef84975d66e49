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
    ap.add_argument("--pass1-groups", type=int, default=None, help="launch groups per set of sampler launches (Trainer.pass1_groups; default 4)")
    ap.add_argument("--framework", choices=["VolSDF", "NeuS"], default="VolSDF", help="NeuS: neus_fangzhou_vangogh.yaml dims, 64 + 64 spp, radiance net frozen (neus.py:455-456)")
    ap.add_argument("--no-keep", action="store_true")
    ap.add_argument("--perturb", action="store_true", help="render_kwargs_train as the reference builds them (perturb=True, volsdf.py:982): pass 2 re-samples")
    ap.add_argument("--second-sampler-run", action="store_true", help="with --perturb: pass 2 runs Algorithm 1 again (Trainer(share_algorithm1=False)) "
                    "instead of taking its samples from pass 1's run")
    ap.add_argument("--no-vgg", action="store_true", help="leave the VGG16 perceptual term (random weights) out of the style loss")
    args = ap.parse_args()
    from nerfart_amd import bench_util
    dev = torch.device("cuda", 0)
    H, W = args.H, args.W
    ctx = bench_util.finetune_setup(dev, H, W, with_vgg=not args.no_vgg, pass2_rays=args.pass2_rays, patches_per_launch=args.patches_per_launch, pass1_groups=args.pass1_groups, framework=args.framework)
    ctx["trainer"].share_algorithm1 = not args.second_sampler_run
    m, loss, eik, _ = bench_util.finetune_steps(ctx, args.steps, warmup=1, keep=not args.no_keep, perturb=args.perturb)
    rkp = dict(ctx["rk"], perturb=args.perturb)
    extra = {"framework": args.framework, "perturb": args.perturb, "pass2_resamples": ctx["trainer"].resamples(rkp), "one_algorithm1_run_for_both_passes": ctx["trainer"].shares_algorithm1(rkp)}
    if extra["pass2_resamples"] and args.framework == "VolSDF":
        extra["pass2_sampler_alone_s"] = round(bench_util.pass2_sampler_seconds(ctx), 3)
    print(json.dumps({**extra, "workload": f"fine-tune step {H}x{W}, {args.framework} {'128+64' if args.framework == 'VolSDF' else '64+64'} spp, CLIP ViT-B/32 + VGG16 random weights", "steps": args.steps, "pass1_state_kept": not args.no_keep, "patches_per_launch": args.patches_per_launch, "vgg_perceptual_term": not args.no_vgg,
                      "s_per_step": round(sum(m), 3), "pass1_render_s": round(m[0], 3), "style_losses_fwd_bwd_s": round(m[1], 3),
                      "pass2_sampler_autograd_s": round(m[2], 3), "adam_s": round(m[3], 4), "loss": loss, "eikonal": eik,
                      "rays_per_s": round(H * W / sum(m), 1), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))


if __name__ == "__main__":
    main()
