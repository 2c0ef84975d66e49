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
