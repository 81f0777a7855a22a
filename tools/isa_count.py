#!/usr/bin/env python3
"""Instruction mix of the hot-loop bodies in build/asm/*.s (made by `make -C htscodecs_amd/csrc asm`).
usage: tools/isa_count.py <kernel-name-substring> [block-label-substring]"""
import re
import sys
from collections import Counter

def blocks_of(src, name):
    i = src.index(name + ':')
    j = src.index('.Lfunc_end', i)
    blocks, cur, lab = [], [], 'entry'
    for l in src[i:j].split('\n'):
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append((lab, cur)); cur = []; lab = l.strip()
        elif l.strip() and not l.strip().startswith(('.', ';')):
            cur.append(l.strip())
    blocks.append((lab, cur))
    return blocks

def main():
    import glob, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'asm')
    want = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    for f in glob.glob(os.path.join(root, '*gfx950.s')):
        src = open(f).read()
        for name in sorted(set(re.findall(r'^(_Z\w+):', src, flags=re.M))):
            if want not in name:
                continue
            bl = [b for b in blocks_of(src, name) if sub in b[0]]
            for lab, b in sorted(bl, key=lambda b: -len(b[1]))[:2]:
                c = Counter(x.split()[0] for x in b)
                cat = lambda p: sum(v for k, v in c.items() if k.startswith(p))
                print(f"{name}\n  {lab[:100]}\n  total {len(b)}  valu {cat('v_')}  ds {cat('ds_')}  salu {cat('s_')}  vmem {cat('global_') + cat('buffer_') + cat('flat_')}")
                print('  ', c.most_common(50))

if __name__ == '__main__':
    main()
