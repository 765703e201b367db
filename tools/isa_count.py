#!/usr/bin/env python3
"""Count instruction classes per kernel in hipcc -S output (tools/isa_count.py file.s [name-filter])."""
import collections
import re
import sys

def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if flt and flt not in name:
            continue
        c = collections.Counter()
        for line in body.split('\n'):
            line = line.strip()
            if not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
                continue
            c[line.split()[0]] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        mul = sum(v for k, v in c.items() if 'mul' in k or 'mad' in k)
        print(f"{name[-60:]} total {tot} valu {valu} mul-class {mul} ds {sum(v for k,v in c.items() if k.startswith('ds_'))} "
              f"global {sum(v for k,v in c.items() if k.startswith('global_'))} salu {sum(v for k,v in c.items() if k.startswith('s_'))}")
        print('   ', ' '.join(f"{k}:{v}" for k, v in c.most_common(24)))

main()
