"""Instruction mix of one kernel in a hipcc -S listing:  python scripts/isa_stats.py file.s name-substring"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if sys.argv[2] not in name:
        continue
    code = body.split('.section')[0]
    ins = [l.strip() for l in code.split('\n') if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
    cnt = lambda pat: sum(1 for i in ins if re.match(pat, i))
    meta = dict(re.findall(r'\.amdhsa_(next_free_vgpr|accum_offset|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size) (\d+)', body))
    print(name[:70], 'instrs', len(ins), 'mfma', cnt('v_mfma'), 'valu', cnt(r'v_(?!mfma)'), 'salu', cnt('s_(?!waitcnt|barrier|cbranch|branch|nop)'),
          'dsread', cnt('ds_read'), 'vmem', cnt('buffer_|global_'), 'waitcnt', cnt('s_waitcnt'), 'scratch', cnt('scratch_'), meta)
