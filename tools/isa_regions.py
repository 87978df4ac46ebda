#!/usr/bin/env python3
"""Static instruction mix of one kernel of a `hipcc --save-temps -gline-tables-only` assembly file per barrier-separated region (in program order), with the
source functions that own each region's VALU instructions.  usage: tools/isa_regions.py file.s kernel_substring"""
import re, sys, collections, bisect, os
path, kern = sys.argv[1], sys.argv[2]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'kvazaar_amd', 'csrc')
files, ranges = {}, {}
def fn_of(fname, line):
    if fname not in ranges:
        st = []
        p = os.path.join(csrc, fname)
        if os.path.exists(p):
            for i, l in enumerate(open(p), 1):
                m = re.match(r'\s*(?:template\s*<[^>]*>\s*)?(?:KVZ_DEV|KVZ_HD|__device__|__global__)[^;(]*?(\w+)\s*\(', l)
                if m and not l.strip().startswith('//'): st.append((i, m.group(1)))
        ranges[fname] = st
    st = ranges[fname]
    k = bisect.bisect_right([a for a, _ in st], line) - 1
    return st[k][1] if k >= 0 else fname
inside, cur, region = False, None, 0
reg = collections.defaultdict(collections.Counter); regfn = collections.defaultdict(collections.Counter)
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m: files[int(m.group(1))] = m.group(3) or m.group(2); continue
    if re.match(r'^_Z\w+:', ln): inside = kern in ln; continue
    if s.startswith('.Lfunc_end'): inside = False
    if not inside: continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if not s or s.startswith('.') or s.startswith(';') or s.endswith(':'): continue
    op = s.split()[0]
    if op == 's_barrier': region += 1; continue
    if op.startswith('v_mfma'): c = 'mfma'
    elif op in ('v_readlane_b32', 'v_writelane_b32'): c = 'lane'
    elif op.startswith('v_'): c = 'valu'
    elif op.startswith('ds_'): c = 'lds'
    elif op.startswith(('global_', 'scratch_', 'flat_', 'buffer_')): c = 'vmem'
    elif op.startswith('s_waitcnt') or op == 's_nop': c = 'wait'
    elif op.startswith('s_'): c = 'salu'
    else: c = 'o'
    reg[region][c] += 1
    if c in ('valu', 'lane') and cur: regfn[region][fn_of(files.get(cur[0], '?'), cur[1])] += 1
tot = collections.Counter()
for r in sorted(reg):
    c = reg[r]; tot.update(c)
    top = ', '.join(f'{k}:{v}' for k, v in regfn[r].most_common(5))
    print(f"R{r:02d} valu {c['valu']:5d} lane {c['lane']:4d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:3d} mfma {c['mfma']:3d} | {top}")
print('total', dict(tot))
