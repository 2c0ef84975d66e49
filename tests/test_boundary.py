"""The product never routes through the oracle or any CPU fallback (task rule 3)."""
import os
import re

from conftest import REPO


def _py_files(root):
    for d, _, fs in os.walk(root):
        if "_build" in d or "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                yield os.path.join(d, f)


def test_package_does_not_import_oracle_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|/root/reference", re.M)
    for root in (os.path.join(REPO, "nerfart_amd"),):
        for p in _py_files(root):
            assert not pat.search(open(p).read()), f"{p} references the oracle / the reference tree"


def test_runtime_files_do_not_read_reference_tree():
    for name in ("bench.py", "__graft_entry__.py"):
        p = os.path.join(REPO, name)
        if os.path.exists(p):
            assert "/root/reference" not in open(p).read()


def test_hip_module_has_no_fallback_branch():
    src = open(os.path.join(REPO, "nerfart_amd", "hip.py")).read()
    assert "raise ImportError" in src and "fallback" in src
    nets = open(os.path.join(REPO, "nerfart_amd", "nets.py")).read()
    assert "F.linear" not in nets and "softplus" not in nets, "nets.py must not carry an eager compute path"


def test_every_cited_profile_exists():
    """DESIGN.md / README.md / INTEGRATION.md cite their evidence as `profiles/<file>` (or a prefix): each citation resolves."""
    import glob
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        for m in set(re.findall(r"profiles/[A-Za-z0-9_.\-\*]+", open(os.path.join(REPO, doc)).read())):
            pat = os.path.join(REPO, m.rstrip(".,;:)"))
            if not glob.glob(pat) and not glob.glob(pat + "*"):
                missing.append((doc, m))
    assert not missing, missing


def _cited_evidence():
    """(document, line number, line, cited file) for every `profiles/<file>` citation of a PMC summary / parity table in the documents and
    docstrings that quote figures from them."""
    import glob
    docs = ["DESIGN.md", "README.md", "INTEGRATION.md", "bench.py", os.path.join("tests", "test_gpu_configs.py"), os.path.join("nerfart_amd", "bench_util.py")]
    out = []
    for doc in docs:
        section_historical = False
        for n, line in enumerate(open(os.path.join(REPO, doc)).read().splitlines(), 1):
            if line.startswith("#") and doc.endswith(".md"):
                section_historical = "historical" in line.lower()
            for m in re.findall(r"profiles/[A-Za-z0-9_.\-\*]+", line):
                m = m.rstrip(".,;:)")
                if not (m.endswith("_pmc_summary.json") or m.endswith("_parity_table.json")):
                    continue
                for f in glob.glob(os.path.join(REPO, m)):
                    out.append((doc, n, line, f, section_historical or "historical" in line.lower()))
    return out


def test_cited_pmc_summaries_and_parity_tables_are_of_these_sources():
    """A figure quoted from `profiles/*_pmc_summary.json` / `*_parity_table.json` must come from the kernel sources in the tree: the file's
    `csrc_sha256` stamp equals hip.csrc_sha256() - unless the citation is marked historical (the word on the citing line, or in the heading of
    the section it sits in: the round-by-round history tables of DESIGN.md)."""
    import json
    from nerfart_amd import hip
    now = hip.csrc_sha256()
    stale = []
    for doc, n, line, f, historical in _cited_evidence():
        if historical:
            continue
        stamp = json.load(open(f)).get("csrc_sha256")
        if stamp != now:
            stale.append(f"{doc}:{n} cites {os.path.relpath(f, REPO)} (csrc {str(stamp)[:12]}..., tree {now[:12]}...)")
    assert not stale, "\n".join(stale)
