"""dev tool: opcode histogram of one kernel from a --save-temps .s file, whole kernel and its hottest loop (the longest
backward-branch region). usage: python tools/isa_hist.py <file.s> <mangled-name-substring>"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(\S+):\s*; @\S+$', s, re.M)]
for k, (pos, name) in enumerate(starts):
    if pat in name:
        end = s.index('s_endpgm', pos)
        body = s[pos:end]
        break
else:
    sys.exit("kernel not found")
print(name)
rows = []  # (label or None, opcode)
for l in body.splitlines():
    t = l.strip()
    if not t or t.startswith((';', '.amd', '.p2', '.sec', '.glob', '.type', '.prot', '.weak')):
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        rows.append((m.group(1), None))
        continue
    if t.startswith('.') or t.endswith(':'):
        continue
    rows.append((None, t.split()[0], t))
ins = [r for r in rows if r[0] is None]
print('instructions in the kernel:', len(ins))
# loops: a branch to a label defined earlier
labpos = {}
loops = []
for i, r in enumerate(rows):
    if r[0]:
        labpos[r[0]] = i
    elif r[1].startswith(('s_cbranch', 's_branch')):
        tgt = r[2].split()[-1]
        if tgt in labpos:
            loops.append((i - labpos[tgt], labpos[tgt], i))
loops.sort(reverse=True)
def hist(lo, hi, title):
    c = collections.Counter(r[1] for r in rows[lo:hi] if r[0] is None)
    n = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    mad = c.get('v_mad_u64_u32', 0)
    print(f'== {title}: {n} instructions, VALU {valu}, v_mad_u64_u32 {mad} ({100.0*mad/max(valu,1):.1f} % of VALU)')
    for k, v in c.most_common(28):
        print(f'   {v:6d} {k}')
hist(0, len(rows), 'whole kernel')
for ln, lo, hi in loops[:2]:
    hist(lo, hi, f'loop of {ln} rows')
