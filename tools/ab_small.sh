#!/bin/bash
# A/B of library builds on small batches (the short-step routes): single 1 MiB block and 1,024 x 1 MiB q40 order 1,
# chain kernels and whole passes.  Variants: htscodecs_amd/variants/lib<name>.so (see tools/ab_variants.sh).
#   gpurun -- 'bash tools/ab_small.sh'        SHAPES overrides the block counts
cd ${GRAFT_REPO_ROOT:-.}

for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then export R4X16_LIB=$PWD/htscodecs_amd/librans4x16_hip.so; else export R4X16_LIB=$PWD/htscodecs_amd/variants/lib$v.so; fi   # (htscodecs_amd/lib.py: the shipped library is never replaced)
  echo "== $v"
  SHAPES=${SHAPES:-1,1024} python3 tools/batch_sweep.py 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'blocks' in d: print('  %6d x %7d %-8s o%-3d  enc %8.3f (chain %8.3f)  dec %8.3f (chain %8.3f)  ok %s %s' % (d['blocks'], d['block_size'], d['data'], d['order'], d['enc_ms'], d['enc_chain_ms'], d['dec_ms'], d['dec_chain_ms'], d['roundtrip_ok'], d['bytes_equal_cpu']))
    else: print('  single call %7d: compress %.3f ms  uncompress %.3f ms' % (d['block_size'], d['compress_ms'], d['uncompress_ms']))
"
done

