"""The hand-counted `s_waitcnt vmcnt(N)` of the Winograd GEMM kernels (conv_wino6.hip: w6_wait / w6_wait_last, conv_wino4.hip: w4_wait / w4_wait_last).

Vector-memory operations of a wave retire in order on one counter, and the compiler does not see the kernel's inline-asm loads, LDS-DMAs and
stores, so every wait inside a position names exactly the operations YOUNGER than the weight group it needs.  Here the
constexpr tables are taken out of the sources as they stand, compiled with g++, and compared with a simulation of a position's issue order:
    group i:  [weight request for stream group i + PF]  [wait for group i's weights]  [slab DMA i, i < 8]  [a result store, groups 8 .. 23 of storing positions]
(the last position of a split job requests no weights beyond its row and fetches no next position).  A count too small would wait for more
than it needs (slower); a count too large would read registers whose load is still in flight -- wrong sums, differently on every run."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <cstdio>
#define PF %(pf)s
#define %(P)s_PF PF
%(tables)s
// ops younger than group i's weight request at the moment of its wait, by simulation of the issue order
static int younger(int i, bool stores, bool lp) {
    // program order of the position's groups 0 .. i (and, for i < PF, the tail of the position before: its groups 25 .. 31 issue nothing but requests)
    int n = 0;
    // requests: the one for stream group g is issued at the top of group g - PF
    for (int g = i + 1; g <= i + PF; g++) if (!lp || g < 32) n++;      // requests for groups i + 1 .. i + PF, all issued by now (the last at the top of group i)
    for (int k = 0; k < i; k++) {      // DMAs and stores of this position's groups k < i issued after group i's request (issued at the top of group i - PF)
        if (k < i - PF) continue;      // older than the request
        if (!lp && k < 8) n++;         // slab DMA k
        if (stores && k >= 8 && k < 24) n++;
    }
    return n;
}
int main() {
    int bad = 0;
    for (int i = 0; i < 32; i++) {
        for (int st = 0; st < 2; st++) {
            const int want = younger(i, st, false), got = %(p)s_wait(i, st);
            if (want != got) { printf("%(p)s_wait(%%d, %%d) = %%d, the issue order says %%d\n", i, st, got, want); bad++; }
        }
        const int want = younger(i, true, true), got = %(p)s_wait_last(i);
        if (want != got) { printf("%(p)s_wait_last(%%d) = %%d, the issue order says %%d\n", i, got, want); bad++; }
    }
    // the wait at a position's start, vmcnt(PF): everything but the PF youngest requests has landed -- the slab DMAs of the position before are older than
    // every request of its groups 8 .. 31
    if (!(32 - 8 >= PF)) { printf("position-start wait does not cover the slab DMAs\n"); bad++; }
    printf("%%d mismatches\n", bad);
    return bad != 0;
}
"""


def tables_of(path, prefix):
    src = open(path).read()
    a = src.index(f"constexpr int {prefix}_store_group")
    b = src.index("}", src.index(f"constexpr int {prefix}_wait_last")) + 1
    b = src.index("\n}\n", src.index(f"constexpr int {prefix}_wait_last")) + 3
    pf = re.search(rf"#define {prefix.upper()}_PF (\d+)", src).group(1)
    return src[a:b], pf


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("fname,prefix", [("conv_wino6.hip", "w6"), ("conv_wino4.hip", "w4")])
def test_wait_counts_match_the_issue_order(tmp_path, fname, prefix):
    tables, pf = tables_of(os.path.join(ROOT, "alphazero-connect6_amd", "csrc", fname), prefix)
    assert "static_assert" in tables      # (the stores-older-than-the-next-position's-waits assertion travels with the tables)
    cpp = tmp_path / "waits.cpp"
    cpp.write_text(HARNESS % dict(pf=pf, P=prefix.upper(), p=prefix, tables=tables))
    exe = tmp_path / "waits"
    subprocess.check_call(["g++", "-std=c++17", "-O1", str(cpp), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("0 mismatches"), out.stdout
