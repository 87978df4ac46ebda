#!/usr/bin/env python3
"""Static instruction mix of one kernel of a `hipcc --save-temps -gline-tables-only` assembly file, per source line / per function of the csrc files.
usage: tools/isa_lines.py file.s kernel_substring [--top N] [--funcs]
Classes: valu (v_*, incl. readlane / writelane / dpp), salu, lds (ds_*), vmem (global_ / scratch_ / flat_ / buffer_), mfma.
Static counts; multiply by a phase's executions per CTU for the dynamic figure."""
import re, sys, collections, bisect, os
path, kern = sys.argv[1], sys.argv[2]
top = 40
if '--top' in sys.argv: top = int(sys.argv[sys.argv.index('--top') + 1])
files = {}
cur = None
inside = False
per_line = collections.defaultdict(lambda: collections.Counter())
tot = collections.Counter()
spill = collections.Counter()
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)
    if m: files[int(m.group(1))] = m.group(3); continue
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"', s)
    if m: files[int(m.group(1))] = m.group(2); continue
    if re.match(r'^_Z\w+:', ln):
        inside = kern in ln
        continue
    if s.startswith('.Lfunc_end'): inside = False
    if not inside: continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if not s or s.startswith('.') or s.startswith(';') or s.endswith(':'): continue
    op = s.split()[0]
    if op.startswith('v_mfma') or op.startswith('v_smfma'): cls = 'mfma'
    elif op.startswith('v_'): cls = 'valu'
    elif op.startswith('ds_'): cls = 'lds'
    elif op.startswith(('global_', 'scratch_', 'flat_', 'buffer_')): cls = 'vmem'
    elif op in ('s_waitcnt', 's_nop'): cls = 'wait'
    elif op == 's_barrier': cls = 'barrier'
    elif op.startswith('s_'): cls = 'salu'
    else: cls = 'other'
    if 'Spill' in ln or 'Reload' in ln: spill[cls] += 1
    if op in ('v_readlane_b32', 'v_writelane_b32') : tot['lane_rw'] += 1
    per_line[cur][cls] += 1
    tot[cls] += 1
print('total', dict(tot), 'spill', dict(spill))
if '--funcs' in sys.argv:
    # function ranges of kvz_ctu.hpp from its own text: "  KVZ_DEV ... name(" lines
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'kvazaar_amd', 'csrc')
    ranges = {}
    for fn in set(files.values()):
        p = os.path.join(src, fn)
        if not os.path.exists(p): continue
        starts = []
        for i, l in enumerate(open(p), 1):
            m = re.match(r'\s*(?:template\s*<[^>]*>\s*)?(?:KVZ_DEV|KVZ_HD|__device__|__global__)[^;(]*?(\w+)\s*\(', l)
            if m and not l.strip().startswith('//'): starts.append((i, m.group(1)))
        ranges[fn] = starts
    agg = collections.defaultdict(collections.Counter)
    for (f, l), c in per_line.items():
        fn = files.get(f, '?')
        st = ranges.get(fn, [])
        k = bisect.bisect_right([a for a, _ in st], l) - 1
        name = st[k][1] if k >= 0 else '?'
        agg[(fn, name)].update(c)
    for (fn, name), c in sorted(agg.items(), key=lambda kv: -kv[1]['valu'])[:top]:
        print(f"{fn:22s} {name:28s} valu {c['valu']:6d} salu {c['salu']:6d} lds {c['lds']:5d} vmem {c['vmem']:5d} mfma {c['mfma']:4d}")
else:
    for (f, l), c in sorted(per_line.items(), key=lambda kv: -kv[1]['valu'])[:top]:
        print(f"{files.get(f,'?')}:{l}  valu {c['valu']} salu {c['salu']} lds {c['lds']} vmem {c['vmem']}")
