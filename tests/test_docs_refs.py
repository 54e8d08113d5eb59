"""The documents cite captures under profiles/ and files of the tree by name: every such name must exist (a `*` or a
`{a,b}` in a cited name is a glob; `<...>` marks a placeholder and is skipped)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["README.md", "DESIGN.md", "BASELINE.md", "INTEGRATION.md", os.path.join("profiles", "README.md")]


def _expand(name):
    # {a,b} alternatives -> several globs
    m = re.search(r"\{([^{}]*)\}", name)
    if not m:
        return [name]
    out = []
    for alt in m.group(1).split(","):
        out += _expand(name[:m.start()] + alt + name[m.end():])
    return out


def test_cited_profile_files_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"profiles/([A-Za-z0-9_.*{},\-]+)", text):
            name = m.group(1).rstrip(".,:;)")
            if "<" in name or name in ("", "README.md") or name.endswith("_"):
                if not name.endswith("_"):
                    continue
                name += "*"                       # "profiles/r03c_" followed by prose
            hits = []
            for g in _expand(name):
                hits += glob.glob(os.path.join(ROOT, "profiles", g)) or glob.glob(os.path.join(ROOT, "profiles", g + "*"))
            if not hits:
                missing.append((doc, name))
    assert not missing, missing


def test_cited_source_files_exist():
    missing = []
    pat = re.compile(r"`((?:tools|tests|marlin_amd|oracle|shim|include|examples)/[A-Za-z0-9_./\-]+\.(?:py|sh|hip|cuh|h|c|rs|json|md|patch|toml))`")
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in pat.finditer(text):
            rel = m.group(1)
            if "arkworks_" in rel:                # written by `cargo test` on a machine with a Rust toolchain
                continue
            if not (os.path.exists(os.path.join(ROOT, rel)) or os.path.exists(os.path.join(ROOT, "shim", rel))):
                missing.append((doc, rel))
    assert not missing, missing
