"""usage: python tools/loopstat.py <file.s> <mangled-kernel-name-substring>  — per loop of the kernel's ISA (backward
branches): lines, vector ALU / spill (v_readlane, v_writelane) / scalar-load / buffer-load / LDS / waitcnt instructions.
Make the .s with: hipcc <flags of orbit_amd/csrc/Makefile> -S --cuda-device-only -o x.s orbit_amd/csrc/<unit>.hip"""
import re
import sys

t = open(sys.argv[1]).read()
names = [m for m in re.findall(r"^(_Z\w+):", t, flags=re.M) if sys.argv[2] in m]
for name in names:
    i = t.index(name + ":")
    j = t.index(".Lfunc_end", i)
    body = t[i:j].split("\n")
    print(name, len(body), "lines")
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < k:
                loops.append((labels[tgt], k))

    def count(seg, pat):
        return sum(1 for x in seg if re.search(pat, x))

    VALU, DS = r"^\s+v_", r"^\s+ds_"

    for a, b in [(0, len(body) - 1)] + loops:
        seg = body[a:b + 1]
        print(f"  {'whole' if a == 0 else 'loop '} {a}-{b}: valu {count(seg, VALU)}, readlane {count(seg, 'v_readlane')}, "
              f"writelane {count(seg, 'v_writelane')}, s_load {count(seg, 's_load')}, buffer_load {count(seg, 'buffer_load')}, "
              f"global_load {count(seg, 'global_load')}, ds {count(seg, DS)}, waitcnt {count(seg, 's_waitcnt')}")
