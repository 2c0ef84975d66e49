"""CLIP ViT-B/32 (rows a23 / B4): image and text encoders with the parameter names and shapes of the OpenAI
checkpoint the reference loads with `clip.load("ViT-B/32")` (criteria/clip_loss.py:165), so `ViT-B-32.pt`'s state
dict loads with `load_state_dict` as is.  Written from the public architecture description:

  image: conv1 768x3x32x32 stride 32, no bias -> [B,49,768] + class token + positional_embedding[50,768] -> ln_pre
         -> 12 x [x += attn(ln_1 x); x += c_proj(quick_gelu(c_fc(ln_2 x)))], 12 heads -> ln_post(x[:,0]) @ proj[768,512]
  text : token_embedding[49408,512] + positional_embedding[77,512] -> 12 causal blocks, 8 heads -> ln_final
         -> x[b, argmax(tokens[b])] @ text_projection[512,512]

On the GPU the weights are fp16 with fp32 LayerNorm statistics, as the reference's `clip.load(..., device="cuda")`.
STATUS: library path - GEMMs / attention through torch (rocBLAS, SDPA); with the text features cached the 16 image
encodes of a fine-tune step are < 1 % of the step's flops.  No tokenizer ships (the BPE vocabulary is
a third-party data file): `encode_text` takes token ids.  PARITY UNPINNED against OpenAI weights (none on disk);
architecture pinned against `transformers.CLIPModel` in tests/test_clip.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm(nn.LayerNorm):
    """fp32 statistics on fp16 activations."""

    def forward(self, x):
        return F.layer_norm(x.float(), self.normalized_shape, self.weight.float(), self.bias.float(), self.eps).to(x.dtype)


class _Attention(nn.Module):
    """Packed-QKV multi-head attention with nn.MultiheadAttention's parameter names."""

    def __init__(self, width: int, heads: int):
        super().__init__()
        self.heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)

    def forward(self, x, causal: bool):
        B, L, C = x.shape
        q, k, v = F.linear(x, self.in_proj_weight, self.in_proj_bias).view(B, L, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        return self.out_proj(o.transpose(1, 2).reshape(B, L, C))


class _MLP(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.c_fc = nn.Linear(width, 4 * width)
        self.c_proj = nn.Linear(4 * width, width)

    def forward(self, x):
        h = self.c_fc(x)
        return self.c_proj(h * torch.sigmoid(1.702 * h))            # QuickGELU


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width: int, heads: int, causal: bool):
        super().__init__()
        self.causal = causal
        self.attn = _Attention(width, heads)
        self.ln_1 = LayerNorm(width)
        self.mlp = _MLP(width)
        self.ln_2 = LayerNorm(width)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x), self.causal)
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, causal: bool):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, causal) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, causal=False)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x):
        # the stride-32 / kernel-32 convolution is a GEMM over non-overlapping patches (no MIOpen: its first use of a new
        # convolution shape compiles kernels for minutes on a fresh box)
        B, C, H, W = x.shape
        ps = self.conv1.kernel_size[0]
        x = x.reshape(B, C, H // ps, ps, W // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // ps) * (W // ps), C * ps * ps)
        x = x @ self.conv1.weight.reshape(self.conv1.out_channels, -1).t()             # [B, 49, width]
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.transformer(self.ln_pre(x))
        return self.ln_post(x[:, 0, :]) @ self.proj


class CLIP(nn.Module):
    """ViT-B/32 by default (the only variant the reference uses)."""

    def __init__(self, embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                 context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12):
        super().__init__()
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads, causal=True)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self._init()

    def _init(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        for tr in (self.transformer, self.visual.transformer):
            width = tr.resblocks[0].ln_1.normalized_shape[0]
            n = len(tr.resblocks)
            for blk in tr.resblocks:
                nn.init.normal_(blk.attn.in_proj_weight, std=width ** -0.5)
                nn.init.normal_(blk.attn.out_proj.weight, std=(width ** -0.5) * ((2 * n) ** -0.5))
                nn.init.normal_(blk.mlp.c_fc.weight, std=(2 * width) ** -0.5)
                nn.init.normal_(blk.mlp.c_proj.weight, std=(width ** -0.5) * ((2 * n) ** -0.5))
        nn.init.normal_(self.text_projection, std=self.transformer.resblocks[0].ln_1.normalized_shape[0] ** -0.5)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype))

    def encode_text(self, tokens):
        x = self.token_embedding(tokens).type(self.dtype) + self.positional_embedding.type(self.dtype)
        x = self.ln_final(self.transformer(x))
        return x[torch.arange(x.shape[0], device=x.device), tokens.argmax(dim=-1)] @ self.text_projection


def build_clip(device="cuda", seed: int = 0, state_dict=None) -> CLIP:
    """Random-weight ViT-B/32 (seeded) or, with `state_dict`, the OpenAI checkpoint; fp16 on the GPU like clip.load."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = CLIP()
    torch.random.set_rng_state(g)
    if state_dict is not None:
        sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        model.load_state_dict(sd)
    model = model.to(device)
    if torch.device(device).type == "cuda":
        model = model.half()
    return model.eval().requires_grad_(False)


def synthetic_tokens(text: str, context_length: int = 77, vocab_size: int = 49408) -> torch.Tensor:
    """Stand-in for clip.tokenize when no BPE vocabulary is on disk: deterministic ids from the string's bytes,
    <start> = vocab-2, <end> = vocab-1 (the arg-max, as in CLIP's vocabulary), zero padded."""
    body = [(b * 7919 + i * 104729) % (vocab_size - 2) for i, b in enumerate(text.encode("utf-8"))][: context_length - 2]
    ids = [vocab_size - 2] + body + [vocab_size - 1]
    return torch.tensor(ids + [0] * (context_length - len(ids)), dtype=torch.long)
