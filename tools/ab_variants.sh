cd $GRAFT_REPO_ROOT
cp htscodecs_amd/librans4x16_hip.so /tmp/base.so
for v in base A B C D E F; do
  if [ $v = base ]; then cp /tmp/base.so htscodecs_amd/librans4x16_hip.so; else cp build/variants/lib$v.so htscodecs_amd/librans4x16_hip.so; fi
  echo "== $v"
  BS=1048576 python3 tools/sweep.py 15360 2>&1 | grep nblk | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('O1 q40  enc', d['enc_chain_ms'], 'dec', d['dec_chain_ms'], 'step', d['step_ms'], d['ok'])"
  BS=1048576 ORDER=0 python3 tools/sweep.py 7680 2>&1 | grep nblk | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('O0 q40  enc', d['enc_chain_ms'], 'dec', d['dec_chain_ms'], 'step', d['step_ms'], d['ok'])"
done
cp /tmp/base.so htscodecs_amd/librans4x16_hip.so
