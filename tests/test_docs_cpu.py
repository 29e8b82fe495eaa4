"""DESIGN.md stays reviewable: lines of at most 200 characters, the sections the task statement asks for, every file it cites under
profiles/ / tests/ / tools/ / csrc/ exists, and every `DESIGN <section>` reference in the sources points at a section the document has."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()


def test_lines_are_short():
    long_lines = [(i + 1, len(l)) for i, l in enumerate(DOC.splitlines()) if len(l) > 200]
    assert not long_lines, long_lines


def test_required_sections_are_there():
    heads = re.findall(r"^#+ (.+)$", DOC, re.M)
    for want in ("0. Status", "1. The path and its boundary", "2. Oracle", "3. C6RNG", "4. Data layout in HBM", "5. Kernels", "6. Round loop", "7. Measurement",
                 "8. Out of scope", "9. Levers"):
        assert any(h.startswith(want) for h in heads), want


def test_cited_files_exist():
    missing = []
    for path in set(re.findall(r"`((?:profiles|tests|tools|docs|oracle|include)/[A-Za-z0-9_./-]+\.[a-z]+)`", DOC)):
        if "*" in path or os.path.exists(os.path.join(ROOT, path)):
            continue
        missing.append(path)
    for name in set(re.findall(r"`(r0[0-9]_[A-Za-z0-9_]+\.(?:json|txt|csv|md))`", DOC)):      # short form inside a list of profiles
        if not os.path.exists(os.path.join(ROOT, "profiles", name)):
            missing.append("profiles/" + name)
    for name in set(re.findall(r"`((?:csrc/)?[a-z0-9_]+\.(?:hip|cpp|inc|h))`", DOC)):
        base = os.path.basename(name)
        if not (glob.glob(os.path.join(ROOT, "alphazero-connect6_amd", "csrc", base)) or glob.glob(os.path.join(ROOT, "include", base))):
            missing.append(name)
    assert not missing, sorted(missing)


def test_section_references_in_the_sources_resolve():
    have = set(re.findall(r"^#+ ([0-9]+[a-z]?(?:\.[0-9]+[a-z]?)?)[. ]", DOC, re.M))
    bad = []
    files = glob.glob(os.path.join(ROOT, "alphazero-connect6_amd", "**", "*.*"), recursive=True) + glob.glob(os.path.join(ROOT, "*.py")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "tests", "*.py"))
    for f in files:
        if not f.endswith((".py", ".hip", ".cpp", ".h", ".inc")) or os.path.basename(f) == "test_docs_cpu.py":
            continue
        for m in re.finditer(r"DESIGN(?:\.md)? (?:§ ?)?([0-9]+[a-z]?(?:\.[0-9]+[a-z]?)?)", open(f, encoding="utf-8", errors="replace").read()):
            if m.group(1) not in have:
                bad.append((os.path.relpath(f, ROOT), m.group(0)))
    assert not bad, bad
