"""Instruction-class string of one kernel from `hipcc -S` output: M mfma, r/w LDS read/write, g/S global load/store, v VALU, s SALU,
[..] s_waitcnt, |B| barrier.  Usage: python tools/isa_classes.py file.s kernel_substring"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if sys.argv[2] in l and re.match(r'^\S+:', l)][0]
end = [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][0] + start
out = []
for l in lines[start + 1:end]:
    t = l.strip()
    if re.match(r'\.LBB\S+:', t):
        out.append('\n' + t.split(':')[0] + ': ')
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith('ds_read') or op.startswith('ds_load'): out.append('r')
    elif op.startswith('ds_write') or op.startswith('ds_store'): out.append('w')
    elif op.startswith('buffer_load') or op.startswith('global_load'): out.append('g')
    elif op.startswith('global_store') or op.startswith('buffer_store'): out.append('S')
    elif op == 's_waitcnt':
        m = re.findall(r'(vmcnt|lgkmcnt)\((\d+)\)', t)
        out.append('[' + ','.join(a[0] + b for a, b in m) + ']')
    elif op.startswith('s_barrier'): out.append('|B|')
    elif op.startswith('v_'): out.append('v')
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): out.append('<br>')
    elif op.startswith('s_nop'): out.append('n')
    elif op.startswith('s_'): out.append('s')
    else: out.append('?')
s = ''.join(out)
s = re.sub(r'v{4,}', lambda m: 'v{%d}' % len(m.group()), s)
s = re.sub(r's{4,}', lambda m: 's{%d}' % len(m.group()), s)
print(s)
