import sys,os
sys.path.insert(0,"tools"); sys.path.insert(0,"."); sys.path.insert(0,"tests")
import batch_sweep as B, htscodecs_amd as H
dc=H.DeviceCodec(0)
for a in ((1024,65536,"mixed",1),(256,1<<20,"mixed",1),(1024,1<<20,"q4",193)):
    r=B.run(dc,*a); print({k:r[k] for k in ("blocks","block_size","data","order","enc_ms","enc_chain_ms","dec_ms","dec_chain_ms","roundtrip_ok","bytes_equal_cpu")})
