"""Where the spilled scalar registers' lane moves sit in a kernel's instruction stream: the loops of the gfx950 assembly (backward branches), innermost first,
with their vector / scalar instruction counts and the v_readlane / v_writelane among them ("660 lane moves of 5 900 static vector
instructions" of c6_search_kernel<2, 2> -- static counts say nothing about how often an instruction runs; the nesting does).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip -c alphazero-connect6_amd/csrc/tree_kernels.hip -o /tmp/tree.o -save-temps=obj
    python tools/isa_loops.py /tmp/tree_kernels-hip-amdgcn-amd-amdhsa-gfx950.s _Z16c6_search_kernelILi2ELi2EEv5C6Devi [min_lines=60]"""
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
min_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 60
text = open(path).read()
body = text[text.index("\n" + sym + ":"):]
body = body[:body.index(".end_amdhsa_kernel")]
lines = body.split("\n")
lab = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = set()
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        loops.add((lab[m.group(1)], i))
is_v = lambda x: re.match(r"\s+v_", x) is not None
is_lane = lambda x: "v_readlane" in x or "v_writelane" in x
tot_v, tot_l = sum(map(is_v, lines)), sum(map(is_lane, lines))
print(f"{sym}: {tot_v} vector instructions, {tot_l} of them v_readlane / v_writelane ({100.0 * tot_l / tot_v:.1f} %)")
# merge loops with the same head (several back edges): keep the widest
heads = {}
for a, b in loops:
    heads[a] = max(heads.get(a, b), b)
rows = sorted(heads.items(), key=lambda ab: ab[1] - ab[0])
print("loops (head line .. last back edge), vector instructions, lane moves among them, lane moves NOT inside a nested loop:")
for a, b in rows:
    if b - a < min_lines:
        continue
    inner = [(x, y) for x, y in rows if a < x and y <= b and (x, y) != (a, b)]
    in_inner = set()
    for x, y in inner:
        in_inner.update(range(x, y + 1))
    v = sum(is_v(lines[i]) for i in range(a, b + 1))
    ln = sum(is_lane(lines[i]) for i in range(a, b + 1))
    own = sum(is_lane(lines[i]) for i in range(a, b + 1) if i not in in_inner)
    print(f"  lines {a:5d} .. {b:5d}  ({b - a + 1:4d})   vector {v:4d}   lane moves {ln:3d}   at this level only {own:3d}")
