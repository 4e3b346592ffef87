"""Static instruction mix between the s_memtime stamps of a -DGUSTO_PROFILE build: python tools/isa_segments.py file.s first_line last_line"""
import sys, collections
path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = open(path).read().split('\n')[a:b]
def kind(op):
    if 'mfma' in op: return 'mfma'
    if op.startswith('v_') and 'f64' in op: return 'f64'
    if 'readlane' in op or 'readfirstlane' in op: return 'rdl'
    if 'writelane' in op: return 'wrl'
    if op.startswith('ds_read'): return 'ds_r'
    if op.startswith('ds_write'): return 'ds_w'
    if op.startswith('ds_'): return 'ds_o'
    if op.startswith('global_load'): return 'g_ld'
    if op.startswith('global_store'): return 'g_st'
    if op.startswith('scratch_load'): return 'sc_ld'
    if op.startswith('scratch_store'): return 'sc_st'
    if 'accvgpr' in op: return 'acc'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
segs = []; cur = collections.Counter(); start = 0
for i, l in enumerate(L):
    t = l.strip()
    if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
    op = t.split()[0]
    if op == 's_memtime':
        segs.append((start, i, cur)); cur = collections.Counter(); start = i
    cur[kind(op)] += 1
segs.append((start, len(L), cur))
keys = ['f64', 'valu', 'acc', 'rdl', 'wrl', 'ds_r', 'ds_w', 'g_ld', 'g_st', 'sc_ld', 'sc_st', 'salu', 'wait', 'br']
print('lines          total ' + ' '.join(f'{k:>5s}' for k in keys))
for s, e, c in segs:
    print(f'{s + a + 1:6d}-{e + a + 1:6d} {sum(c.values()):5d} ' + ' '.join(f'{c[k]:5d}' for k in keys))
