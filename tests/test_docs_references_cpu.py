"""DESIGN.md / INTEGRATION.md / profiles/README.md cite evidence by file name: every cited profiles/, tools/, tests/, oracle/, include/ and package
path must exist in the tree (a renamed or dropped file otherwise leaves a claim without its source)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "INTEGRATION.md", os.path.join("profiles", "README.md"), os.path.join("profiles", "r03_notes_attention_power.md"))


def _missing():
    out = []
    for doc in DOCS:
        txt = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"profiles/([A-Za-z0-9_\-./*]+)", txt):
            f = m.group(1).rstrip(".,;:)")
            if "*" not in f and not os.path.exists(os.path.join(ROOT, "profiles", f)):
                out.append((doc, "profiles/" + f))
        for m in re.finditer(r"`(r0\d_[A-Za-z0-9_\-.]+)`", txt):  # bare round-prefixed names in backticks are profile files
            f = m.group(1)
            if "*" not in f and not os.path.exists(os.path.join(ROOT, "profiles", f)):
                out.append((doc, f))
        for m in re.finditer(r"((?:tools|tests|oracle|ai-toolkit_amd|include)/[A-Za-z0-9_\-./]+\.(?:py|hip|h|sh|md|json|safetensors))", txt):
            if not os.path.exists(os.path.join(ROOT, m.group(1))):
                out.append((doc, m.group(1)))
    return out


def test_every_cited_file_exists():
    assert _missing() == []
