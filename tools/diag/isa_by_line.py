"""Scratch diagnostic: static instruction mix of k_leap_v2<false> per source-line range (needs -gline-tables-only .loc directives).
usage: python tools/isa_by_line.py build/isa/v2.s  (ranges = the phase markers found in the source)"""
import re, sys, collections
src = open('judo_amd/csrc/jh_engine_v2.hip').read().split('\n')
marks = [(i + 1, l.strip()[:70]) for i, l in enumerate(src) if re.search(r'// (=====|---- \()', l)]
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and 'k_leap_v2ILb0' in l][0]
end = [i for i, l in enumerate(lines) if l.startswith('_ZN') and 'k_leap_v2ILb1' in l][0]
files = {}
cur = (None, 0); cnt = collections.defaultdict(collections.Counter)
def cat(op):
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('scratch'): return 'scratch'
    if 'dpp' in op: return 'dpp'
    if op.startswith('v_mov') or op.startswith('v_pk_mov'): return 'mov'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log|div)', op): return 'trans'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'mem'
for l in lines[start:end]:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m or l.strip().startswith('.') or l.strip().startswith(';'): continue
    cnt[cur][cat(m.group(1))] += 1
# file id of the main source = the one with most instructions at lines > 300
byfile = collections.Counter()
for (f, ln), c in cnt.items(): byfile[f] += sum(c.values())
print('instructions per .file id:', dict(byfile))
main = max((f for f in byfile), key=lambda f: sum(sum(c.values()) for (ff, ln), c in cnt.items() if ff == f and ln > 330))
cats = ['valu', 'dpp', 'mov', 'cnd', 'acc', 'scratch', 'lane', 'trans', 'lds', 'salu', 'branch', 'wait', 'mem']
print(f'{"phase (first source line)":72s}' + ''.join(c[:6].rjust(7) for c in cats) + '  total')
bounds = [m[0] for m in marks] + [10 ** 9]
agg = collections.defaultdict(collections.Counter)
for (f, ln), c in cnt.items():
    if f != main: agg[(0, 'inlined helpers / other files')] += c; continue
    k = max([i for i, b in enumerate(bounds[:-1]) if b <= ln], default=None)
    agg[(marks[k][0], marks[k][1]) if k is not None else (1, 'before first marker')] += c
for key in sorted(agg):
    c = agg[key]
    print(f'{key[0]:4d} {key[1]:67s}' + ''.join(str(c[x]).rjust(7) for x in cats) + f'  {sum(c.values())}')
