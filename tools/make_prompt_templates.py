#!/usr/bin/env python
"""Writes nerfart_amd/data/prompt_templates.txt: the 79 prompt templates the reference's style losses average text
features over (the list `imagenet_templates` in criteria/clip_loss.py:6-87, contrastive_loss.py, patchnce_loss.py -
OpenAI CLIP's published ImageNet prompt-engineering list).  DATA, one template per line; read where the reference
lies, build container only:    python tools/make_prompt_templates.py
"""
import ast
import os

SRC = "/root/reference/criteria/clip_loss.py"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerfart_amd", "data", "prompt_templates.txt")

tree = ast.parse(open(SRC).read())
templates = next(ast.literal_eval(n.value) for n in tree.body
                 if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "imagenet_templates")
assert len(templates) == 79 and all("{}" in t for t in templates)
with open(DST, "w") as f:
    f.write("\n".join(templates) + "\n")
print(DST, len(templates))
