// r4x16_enc_step.h - the encoder's per-symbol arithmetic, shared by the chain kernels (r4x16_enc_chain.hip) and the
// rANS 4x8 coder (r4x16_encode.hip).
#pragma once
#include "r4x16_dev.h"

// A prefetched symbol is two words: rcp = rcptab[freq] and sf = start | freq << 16.
// RansEncPutSymbol (rANS_word.h:281-321) with RansEncSymbolInit's parameters (:190-266) derived on
// the fly: x_max = freq << (31 - bits); q = (x * rcp) >> (32 + ceil(log2 freq) - 1) is the exact
// quotient x / freq for freq >= 2 (Alverson), and freq == 1 takes q = x; then
// x' = x + start + q * (M - freq)  ==  ((x / freq) << bits) + x % freq + start.
__device__ __forceinline__ bool enc_wants_emit(u32 x, u32 sf, u32 bits)
{
    return x >= ((sf >> 16) << (31 - bits));
}
__device__ __forceinline__ u32 enc_rcp(gcu32 *rcptab, u32 f)
{
    return rcptab[f < RCPTAB_ENTRIES ? f : 0u];          // idle lanes may hold garbage: stay inside the table
}
__device__ __forceinline__ u32 enc_advance(u32 x, u32 rcp, u32 sf, u32 bits)
{
    const u32 f = sf >> 16, start = sf & 0xffffu;
    const u32 rsh = 31u - (u32)__clz((int)(f - 1u));     // ceil(log2 f) - 1 for f >= 2
    u32 q = __umulhi(x, rcp) >> (rsh & 31u);
    q = f < 2u ? x : q;
    return x + start + q * ((1u << bits) - f);
}

