#!/usr/bin/env python
"""Merge the per-pass PMC text summaries (tools/prof_summary.py output) into one JSON: averages per dispatch per
kernel, plus derived hbm_bytes_corrected_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction,
MI355X_MICROARCH.md HBM section) and mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CYCLES)... see below.
usage: pmc_summary.py out.json pass1.txt pass2.txt ..."""
import json
import re
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    kern = {}
    commands = set()
    for f in files:
        for line in open(f):
            if line.startswith("#") and ": python " in line:
                commands.add("python " + line.split(": python ", 1)[1].strip())
            m = re.match(r"\s*(\d+)\s+([0-9.e+\-]+)\s+([0-9.e+\-]+)\s+(\S+)\s+nerfart::(?:b16::|wgrad::|style::|vgg::|clip::|gemm32::)?(\S+)", line)
            if not m:
                continue
            n, _sum, avg, counter, name = m.groups()
            kern.setdefault(name, {})[counter] = float(avg)
            kern[name].setdefault("dispatches", int(n))
    for name, k in kern.items():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["hbm_bytes_corrected_per_launch"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs:
        # busy fraction = MFMA_BUSY / (1024 SIMDs * GUI_ACTIVE / 8)
        if k.get("GRBM_GUI_ACTIVE"):
            k["mfma_util"] = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0)
    note = ("rocprofv3 --pmc passes of the command(s) in `commands` (1x MI355X); averages per dispatch. "
            "FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
            "(gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md).")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from nerfart_amd import hip
        sha = hip.csrc_sha256()
    except Exception:
        sha = None
    try:                                    # the sampler's launches are per chunk of rays: per-launch figures scale with the default chunk
        from nerfart_amd import volsdf
        chunk = volsdf.DEFAULT_RAYSCHUNK
    except Exception:
        chunk = None
    json.dump({"kernels": kern, "note": note, "csrc_sha256": sha, "default_rayschunk": chunk, "commands": sorted(commands)}, open(out, "w"), indent=1, sort_keys=True)
    print(out, len(kern), "kernels")


if __name__ == "__main__":
    main()
