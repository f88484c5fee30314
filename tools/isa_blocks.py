#!/usr/bin/env python3
"""Instruction counts per basic block of one kernel in a hipcc -S listing.
usage: tools/isa_blocks.py <file.s> <kernel-name-substring> [min-instructions]"""
import re, sys
path, sub = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and sub in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur = [], ['entry', 0, 0, 0, 0, 0]
blocks.append(cur)
for ln in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        cur = [m.group(1), 0, 0, 0, 0, 0]
        blocks.append(cur)
        continue
    t = ln.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    if op.startswith('v_'):
        cur[1] += 1
        if 'f64' in op:
            cur[5] += 1
    elif op.startswith('ds_'):
        cur[2] += 1
    elif op.startswith('s_'):
        cur[3] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        cur[4] += 1
print(f"{lines[start].split(':')[0]}: {len(blocks)} blocks;  [label, VALU, LDS, SALU, VMEM, of-VALU-f64]")
for b in blocks:
    if b[1] + b[2] + b[4] >= minn:
        print('  ', b)
for l in lines[end:end + 60]:
    if any(k in l for k in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize', 'codeLenInByte')):
        print(l.strip())
