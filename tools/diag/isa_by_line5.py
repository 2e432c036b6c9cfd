"""Static instruction mix of one kernel of an assembly listing per source-line range (needs -gline-tables-only .loc directives).
usage: python tools/diag/isa_by_line5.py build/isa/v5.s judo_amd/csrc/jh_engine_v5.hip k_leap_v5ILb0ELi4ELb1 [--lines]"""
import re, sys, collections
asm, srcf, key = sys.argv[1:4]
perline = '--lines' in sys.argv
src = open(srcf).read().split('\n')
marks = [(i + 1, l.strip()[:70]) for i, l in enumerate(src) if re.search(r'// (=====|---- \()', l)]
lines = open(asm).read().split('\n')
fstarts = [i for i, l in enumerate(lines) if l.startswith('_Z') and l.rstrip().endswith(':') or (l.startswith('_Z') and '; @' in l)]
start = [i for i in fstarts if key in lines[i]][0]
end = min([i for i in fstarts if i > start] + [len(lines)])
def cat(op):
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('scratch'): return 'scratch'
    if op.startswith('v_mov') or op.startswith('v_pk_mov'): return 'mov'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log|div)', op): return 'trans'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if 'saveexec' in op or 'exec' in op: return 'exec'
    if op.startswith('s_'): return 'salu'
    return 'mem'
cur = (None, 0); cnt = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+([a-z_0-9]+)(.*)', l)
    if not m or l.strip().startswith('.') or l.strip().startswith(';'): continue
    c = cat(m.group(1))
    if 'dpp' in m.group(2) and c == 'valu': c = 'dpp'
    if 'exec' in m.group(2) and c == 'salu': c = 'exec'
    cnt[cur][c] += 1
byfile = collections.Counter()
for (f, ln), c in cnt.items(): byfile[f] += sum(c.values())
main = max((f for f in byfile), key=lambda f: sum(sum(c.values()) for (ff, ln), c in cnt.items() if ff == f and ln > 330))
cats = ['valu', 'dpp', 'mov', 'cnd', 'cmp', 'scratch', 'lane', 'trans', 'lds', 'salu', 'exec', 'branch', 'wait', 'mem']
print('instructions per .file id:', dict(byfile), 'main', main)
print(f'{"phase (first source line)":72s}' + ''.join(c[:6].rjust(7) for c in cats) + '  total')
bounds = [m[0] for m in marks] + [10 ** 9]
agg = collections.defaultdict(collections.Counter)
for (f, ln), c in cnt.items():
    if f != main: agg[(0, 'inlined helpers / other files (file %d)' % f)] += c; continue
    if perline: agg[(ln, src[ln - 1].strip()[:66])] += c; continue
    k = max([i for i, b in enumerate(bounds[:-1]) if b <= ln], default=None)
    agg[(marks[k][0], marks[k][1]) if k is not None else (1, 'before first marker')] += c
tot = collections.Counter()
for key_ in sorted(agg):
    c = agg[key_]; tot += c
    if perline and sum(c.values()) < 40: continue
    print(f'{key_[0]:4d} {key_[1]:67s}' + ''.join(str(c[x]).rjust(7) for x in cats) + f'  {sum(c.values())}')
print(f'{"":4s} {"TOTAL":67s}' + ''.join(str(tot[x]).rjust(7) for x in cats) + f'  {sum(tot.values())}')
