"""The in-tree build's dependency list: every file a kernel source includes from csrc/ (or include/) must be among the Makefile's HDRS, or an
edit of it would not rebuild libconnect6_amd.so (the library that travels to the GPU box is the one built here); and every file the convolution
kernels of the committed counter summary are compiled from must be in bench.CONV_SOURCES' hash (r5: the GEMM body moved into an include)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alphazero-connect6_amd", "csrc")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def local_includes(path):
    return set(re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), re.M))


def test_makefile_lists_every_local_include():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    hdrs = set(os.path.basename(h) for h in re.search(r"^HDRS := (.*)$", mk, re.M).group(1).split())
    srcs = re.search(r"^SRCS := (.*)$", mk, re.M).group(1).split()
    assert srcs and all(os.path.exists(os.path.join(CSRC, f)) for f in srcs)
    seen, todo = set(), list(srcs)
    while todo:      # (includes of includes too)
        f = todo.pop()
        for inc in local_includes(os.path.join(CSRC, f)):
            b = os.path.basename(inc)
            if b in seen:
                continue
            seen.add(b)
            assert b in hdrs or b in srcs, f"{f} includes {inc}, which the Makefile's HDRS do not list"
            p = os.path.normpath(os.path.join(CSRC, inc))
            if os.path.exists(p) and os.path.dirname(p) == CSRC:
                todo.append(b)
    assert "conv_wino6_gemm.inc" in seen and "c6_wino.h" in seen


def test_conv_sources_hash_covers_what_conv_wino6_is_compiled_from():
    inc = {os.path.basename(i) for i in local_includes(os.path.join(CSRC, "conv_wino6.hip"))}
    kernel_text = {i for i in inc if i.endswith(".inc") or i == "c6_wino.h"}
    assert kernel_text <= set(bench.CONV_SOURCES), (kernel_text, bench.CONV_SOURCES)
    # (and the hash does read those files: it changes with the list)
    assert bench.kernel_source_hash(bench.CONV_SOURCES) != bench.kernel_source_hash(tuple(f for f in bench.CONV_SOURCES if not f.endswith(".inc")))
