"""Basic blocks of one kernel in a hipcc -S listing with their instruction mix: python tools/isa_blocks.py file.s kernel_substr [min_instrs]"""
import re, sys, collections
path, ker = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
L = open(path).read().split('\n')
start = [i for i, l in enumerate(L) if l.startswith('_ZN') and ker in l and ':' in l][0]
end = [i for i, l in enumerate(L) if i > start and l.strip().startswith('.end_amdhsa_kernel')][0]
cur = start
def stats(a, b):
    c = collections.Counter(); br = []
    for l in L[a:b]:
        t = l.strip()
        if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
        op = t.split()[0]; c['n'] += 1
        if 'rsq_f64' in op: c['rsq'] += 1
        if op.startswith('scratch_'): c['scr'] += 1
        if 'accvgpr' in op: c['acc'] += 1
        if op.startswith('ds_'): c['ds'] += 1
        if 'readlane' in op: c['rdl'] += 1
        if op.startswith('s_waitcnt'): c['wait'] += 1
        if 'vmcnt(0)' in t: c['vm0'] += 1
        if op.startswith('v_') and 'f64' in op: c['f64'] += 1
        if op.startswith('global_'): c['glb'] += 1
        if op.startswith('s_cbranch') or op.startswith('s_branch'): br.append(t.split()[-1])
    return c, br
for i in range(start, end + 1):
    if re.match(r'^\.LBB\d+_\d+:', L[i]) or i == end:
        c, br = stats(cur, i)
        if c['n'] >= minn: print(L[cur][:14].ljust(14), cur, i, dict(c), br)
        cur = i
