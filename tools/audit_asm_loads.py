#!/usr/bin/env python
"""Audit of the hand-pipelined LDS fragment reads in the split-bf16 kernels (cdna_hip_programming.md 5.7, item 1: an asm
load's destination counts as written at the end of the asm statement - hipcc may copy / spill / reuse it before the data
lands).  For every inline-asm `ds_read_b128 vA, ...` the first later instruction that touches vA must come after an
`s_waitcnt lgkmcnt(c)` with c <= the number of LDS instructions issued after that read and before the wait.
Straight-line only: a branch or label between a read and its first use is reported.
    usage: python tools/audit_asm_loads.py            (compiles csrc/mlp_*_bf16.hip to ISA and checks every kernel)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerfart_amd", "csrc")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(lines, kernel):
    problems, checked = [], 0
    code = [(i, l.strip()) for i, l in enumerate(lines) if l.strip() and not l.strip().startswith(";")]
    idx_of = {i: k for k, (i, _) in enumerate(code)}
    in_asm = False
    asm_reads = []
    for i, l in enumerate(lines):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
        elif s.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and s.startswith("ds_read_b128"):
            asm_reads.append(i)
    for i in asm_reads:
        dst = regs_of(lines[i].split(",")[0])
        k = idx_of[i]
        lds_after, last_wait_ok = 0, False
        for (j, s) in code[k + 1: k + 4000]:
            if s.endswith(":") or s.startswith("s_cbranch") or s.startswith("s_branch") or s.startswith("s_endpgm"):
                if not last_wait_ok:
                    problems.append((kernel, i, "control flow before the wait", s))
                break
            m = re.match(r"s_waitcnt .*lgkmcnt\((\d+)\)", s)
            if m:
                if int(m.group(1)) <= lds_after:
                    last_wait_ok = True
                continue
            if s.startswith("s_waitcnt") and "lgkmcnt" not in s:
                continue
            touched = regs_of(s.split(";")[0]) & dst
            if s.startswith("ds_") or s.startswith("s_load") or s.startswith("s_buffer_load"):
                lds_after += 1
                if s.startswith("s_"):
                    problems.append((kernel, i, "scalar memory load inside the counted window (out-of-order lgkmcnt)", s))
            if touched:
                checked += 1
                if not last_wait_ok:
                    problems.append((kernel, i, "destination touched before its wait", s))
                break
    return checked, problems


def audit_mfma(lines, kernel, need=12):
    """The inline-asm MFMA triples: hipcc does not pad the MFMA-result -> non-MFMA reader/writer hazard for instructions it
    cannot see (cdna_hip_programming.md 5.7 item 2): at least `need` issue slots (s_nop n counts n + 1) must separate the last
    MFMA of an asm statement from the first other instruction that touches its destination."""
    problems, checked = [], 0
    code = [(i, l.strip()) for i, l in enumerate(lines) if l.strip() and not l.strip().startswith(";")]
    for k, (i, s) in enumerate(code):
        if not s.startswith("v_mfma") or (k + 1 < len(code) and code[k + 1][1].startswith("v_mfma")):
            continue
        dst = regs_of(s.split(",")[0])
        states = 0
        for (j, t) in code[k + 1: k + 200]:
            if t.endswith(":"):
                continue
            touched = regs_of(t.split(";")[0]) & dst
            if touched and not t.startswith("v_mfma"):
                checked += 1
                if states < need:
                    problems.append((kernel, i, f"MFMA result touched after {states} states", t))
                break
            if touched and t.startswith("v_mfma"):
                checked += 1
                break
            m = re.match(r"s_nop (\d+)", t)
            states += (int(m.group(1)) + 1) if m else 1
    return checked, problems


def main():
    total, bad = 0, []
    for src in ("mlp_chain_bf16.hip", "mlp_grad_bf16.hip", "mlp_backward_bf16.hip", "mlp_chain_f16x1.hip"):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-S", "--cuda-device-only",
                                   os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read().split("\n")
        starts = [i for i, l in enumerate(text) if re.match(r"^_ZN7nerfart(3b16|5f16x1)\w+:", l)]
        for a, b in zip(starts, starts[1:] + [len(text)]):
            name = text[a].rstrip(":")
            n, p = audit(text[a:b], name)
            n2, p2 = audit_mfma(text[a:b], name)
            print(f"{src:26s} {name[:60]:60s} asm fragment reads checked: {n:5d}  MFMA results: {n2:5d}  problems: {len(p) + len(p2)}")
            total += n
            bad += p + p2
    for k, i, what, s in bad[:40]:
        print("PROBLEM", k[:50], "line", i, what, "|", s)
    print(f"total reads checked {total}, problems {len(bad)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
