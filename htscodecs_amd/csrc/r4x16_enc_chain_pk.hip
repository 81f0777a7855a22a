// =============================================================================================
// r4x16_enc_chain_pk.hip - k_enc_chain<true, true>: order-1 streams with packed rows (10-bit tables of 20..64 symbols,
// the quality alphabets).  A translation unit of its own because it is compiled with another instruction scheduler
// than the rest of the library, -mllvm -amdgpu-sched-strategy=max-ilp (Makefile): this software-pipelined loop runs 6 %
// faster with it (55.0 -> 51.7 ms on the headline batch), the order-0 pipeline 4.6 % slower (34.6 -> 36.2 ms for
// 15,360 x 1 MiB) and the decoder's dependent chains 5 % slower.
// =============================================================================================
#include "r4x16_enc_chain.h"

extern "C" void r4x16_enc_chain_pk_lds_limit(int bytes)
{
    (void)hipFuncSetAttribute((const void *)k_enc_chain<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
extern "C" const void *r4x16_enc_chain_pk_kernel(void) { return (const void *)k_enc_chain<true, true>; }
