"""The documents cite files (profiles, tools, tests, sources) as evidence: every path they name must exist in the tree."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "profiles/EXPERIMENTS.md", "tools/archive/README.md", "oracle/ref_spv/README.md"]
# `path` tokens rooted in one of the repository's own directories (patterns with *, {a,b} or <tag> are expanded / skipped below)
TOKEN = re.compile(r"`((?:profiles|tools|tests|oracle|include|rvpt_amd)/[A-Za-z0-9_./{},*<>-]+)`")
BARE_PROFILE = re.compile(r"`(r0\d_[A-Za-z0-9_.{},*-]+|pmc_traffic\.json)`")  # profiles/README.md names its neighbours without the directory


def expand(token: str):
    m = re.search(r"\{([^{}]*)\}", token)
    if not m:
        return [token]
    out = []
    for alt in m.group(1).split(","):
        out += expand(token[:m.start()] + alt + token[m.end():])
    return out


@pytest.mark.parametrize("doc", DOCS)
def test_cited_paths_exist(doc):
    text = (ROOT / doc).read_text()
    missing = []
    tokens = set(TOKEN.findall(text))
    if doc.startswith("profiles/"):
        tokens |= {"profiles/" + t for t in BARE_PROFILE.findall(text)}
    for token in sorted(tokens):
        token = token.split("::")[0].rstrip(".,:")
        if "<" in token:  # a placeholder such as profiles/<round>_<tag>_pmc.json
            continue
        for path in expand(token):
            if path.startswith("oracle/_ref") or path.startswith("rvpt_amd/bin") or path.endswith(".so"):
                continue  # build outputs (git-ignored)
            if "*" not in path and not (ROOT / path).exists() and "." not in Path(path).name:
                path += "*"  # a tag such as profiles/r01_hf_bvh names the files that start with it
            hits = list(ROOT.glob(path)) if "*" in path else ([ROOT / path] if (ROOT / path).exists() else [])
            if not hits:
                missing.append(path)
    assert not missing, f"{doc} cites paths that do not exist: {missing}"


@pytest.mark.parametrize("doc", DOCS)
def test_no_placeholder_is_left_unfilled(doc):
    """Numbers that come from the round's last GPU passes are written as @NAME@ while the text is drafted (round 6's DESIGN.md was committed with eleven of them)."""
    left = sorted(set(re.findall(r"@[A-Z][A-Z0-9_]*@", (ROOT / doc).read_text())))
    assert not left, f"{doc}: unfilled placeholders {left}"


LAUNCH_MS = re.compile(r"`(profiles/r\d\d[a-z]?_[A-Za-z0-9_]+_trace_kernel_stats\.csv)`:\s*([0-9]+\.[0-9]+)\s*ms per (?:lone )?(?:\d+-frame )?launch")


@pytest.mark.parametrize("doc", ["DESIGN.md", "README.md", "profiles/README.md"])
def test_quoted_launch_durations_are_the_committed_ones(doc):
    """Every "`profiles/rNN_<tag>_trace_kernel_stats.csv`: X ms per launch" in the documents is the AverageNs of that file's first (dominant) kernel row, to the
    digits quoted (DESIGN.md quoted 0.941 ms for a file that said 0.9546 after a re-take)."""
    import csv
    text = " ".join((ROOT / doc).read_text().split())
    found = LAUNCH_MS.findall(text)
    if doc == "DESIGN.md":
        assert len(found) >= 3, "DESIGN.md quotes the headline's, C3's and C4's launch durations in the checked form"
    for path, quoted in found:
        with open(ROOT / path) as f:
            row = next(csv.DictReader(f))
        avg_ms = float(row["AverageNs"]) * 1e-6
        digits = len(quoted.split(".")[1])
        assert abs(avg_ms - float(quoted)) <= 0.51 * 10 ** -digits, f"{doc}: {path} says {avg_ms:.4f} ms per launch, the text {quoted}"
