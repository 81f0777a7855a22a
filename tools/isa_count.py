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

HOT = {   # kernel-name mangled prefix -> (label, instruction that occurs once per step in the hot body)
    "_Z11k_dec_chainILb1ELi1ELi8EE": ("k_dec_chain<true,1>", "v_lshrrev_b64"),
    "_Z11k_dec_chainILb1ELi6ELi8EE": ("k_dec_chain<true,6>", "v_lshrrev_b64"),
    "_Z11k_enc_chainILb1ELb1EE": ("k_enc_chain<true,true>", "ds_write_b16"),
    "_Z15k_enc_chain_rec": ("k_enc_chain_rec", "ds_write_b16"),
}


def to_json(out_path):
    """Instructions per step of the hot loop bodies (the largest block of each chain kernel that has no per-lane
    liveness selects, i.e. the body full trips take), keyed to the hash of the library built from the same sources:
    what bench.py's roofline.issue is computed from."""
    import glob, os, json, hashlib
    top = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = os.path.join(top, 'build', 'asm')
    with open(os.path.join(top, 'htscodecs_amd', 'librans4x16_hip.so'), 'rb') as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    res = {"library_sha256_16": sha, "how": "make -C htscodecs_amd/csrc asm && python tools/isa_count.py --json <out>", "kernels": {}}
    for f in glob.glob(os.path.join(root, '*gfx950.s')):
        src = open(f).read()
        for name in sorted(set(re.findall(r'^(_Z\w+):', src, flags=re.M))):
            for pre, (label, marker) in HOT.items():
                if not name.startswith(pre):
                    continue
                best = None
                for lab, b in blocks_of(src, name):
                    steps = sum(1 for x in b if x.split()[0] == marker)
                    if steps < 4:
                        continue
                    per = len(b) / steps
                    if best is None or per < best[0]:            # the leanest body with >= 4 steps: the full-trip one
                        c = Counter(x.split()[0] for x in b)
                        cat = lambda p: sum(v for k, v in c.items() if k.startswith(p))
                        best = (per, {"steps_in_block": steps, "instructions_per_step": round(per, 1),
                                      "valu_per_step": round(cat('v_') / steps, 1), "lds_per_step": round(cat('ds_') / steps, 1),
                                      "salu_and_waits_per_step": round(cat('s_') / steps, 1)})
                if best:
                    res["kernels"][label] = best[1]
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


def main():
    import glob, os
    if len(sys.argv) > 2 and sys.argv[1] == '--json':
        return to_json(sys.argv[2])
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
