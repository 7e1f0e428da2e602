#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc --save-temps assembly file:  python tools/isa_loops.py file.s <mangled-name-substring> [dump_from dump_to]"""
import re
import sys
txt = open(sys.argv[1]).read()
names = re.findall(r"^(_Z\w+):", txt, flags=re.M)
name = [n for n in names if sys.argv[2] in n][0]
i = txt.index("\n" + name + ":")
body = txt[i:txt.index("s_endpgm", i)]
lines = []
for l in body.split("\n"):
    t = l.split(";")[0].strip()
    if not t or (t.startswith(".") and not t.endswith(":")):
        continue
    lines.append(t)
labels = {l[:-1]: n for n, l in enumerate(lines) if l.endswith(":")}
loops = []
for n, l in enumerate(lines):
    m = re.match(r"s_cbranch_\w+ (\S+)", l) or re.match(r"s_branch (\S+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        loops.append((n - labels[m.group(1)], labels[m.group(1)], n))
print(name, len(lines), "instructions")
for size, a, b in sorted(loops, reverse=True)[:10]:
    seg = lines[a:b + 1]
    c = lambda p: sum(1 for l in seg if l.startswith(p))
    print(f"loop [{a},{b}] size {size}: valu {c('v_')} salu {c('s_')} ds {c('ds_')} vmem {c('global_') + c('buffer_') + c('flat_')}")
if len(sys.argv) > 4:
    print("\n".join(lines[int(sys.argv[3]):int(sys.argv[4])]))
