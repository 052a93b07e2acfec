"""Print the instruction-class sequence of one kernel from a `hipcc -S --cuda-device-only` listing (diagnostic).

usage: python tools/isa_schedule.py build_mb/loss.s _Z19vocab_stream_kernelILi10ELi0ELi1EE [width]
M = MFMA, e = v_exp, c = v_cvt_pk_bf16, d = ds_read, G = global_load_lds, w = s_waitcnt, B = s_barrier, v = other VALU,
| = branch, . = anything else.
"""
import re
import sys


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    s = open(path).read()
    m = re.search(r'^(' + re.escape(prefix) + r'\w*):[^\n]*\n(.*?)\n\.Lfunc_end\d+:', s, re.S | re.M)
    name, body = m.group(1), m.group(2).split('\n')
    meta = s[s.index('.name:           ' + name) - 2500:s.index('.name:           ' + name) + 2500]
    for key in ('vgpr_count', 'agpr_count', 'sgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count'):
        mm = re.search(r'\.' + key + r':\s+(\d+)', meta)
        print(key, mm.group(1) if mm else None)
    seq = []
    for l in body:
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'):
            if l.endswith(':'):
                seq.append('\n' + l + ' ')
            continue
        op = l.split()[0]
        seq.append('M' if op.startswith('v_mfma') else 'e' if op.startswith('v_exp') else
                   'c' if op.startswith('v_cvt_pk_bf16') else 'd' if op.startswith('ds_read') else
                   'G' if op.startswith('global_load_lds') else 'w' if op.startswith('s_waitcnt') else
                   'B' if op.startswith('s_barrier') else 'v' if op.startswith('v_') else
                   '|' if op.startswith(('s_cbranch', 's_branch')) else '.')
    print(''.join(seq))


if __name__ == '__main__':
    main()
