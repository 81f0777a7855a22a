// r4x16_host.h - what the two host-side translation units of librans4x16_hip.so share: the context, the error
// macro and the entry points of the host-buffer batch machinery (r4x16_host.hip) used by the C ABI (r4x16_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <string>
#include <vector>
#include <mutex>
#include <atomic>
#include <thread>
#include <chrono>
#include <algorithm>

#include "../../include/rans4x16_hip.h"
#include "../../include/rans4x8_hip.h"
#include "r4x16_dev.h"
#include "r4x16_sched.h"

extern "C" {
void r4x16_launch_dec_front(const BatchArgs *, const DecWs *, int, int, hipStream_t, const R4Opts *);
void r4x16_launch_dec_chain(const DecWs *, int, hipStream_t, const R4Fork *, const R4Opts *, SchedHint *);
void r4x16_launch_dec_back(const BatchArgs *, const DecWs *, int, int, hipStream_t, const R4Opts *);
void r4x16_launch_enc_front(const BatchArgs *, const EncWs *, int, int, hipStream_t, const R4Opts *);
void r4x16_launch_enc_tables(const BatchArgs *, const EncWs *, int, int, hipStream_t);
void r4x16_launch_enc_chain(const EncWs *, int, hipStream_t, const R4Fork *, const R4Opts *, SchedHint *);
void r4x16_launch_enc_finish(const BatchArgs *, const EncWs *, int, int, hipStream_t);
u32  r4x16_compress_bound(u32 size, int order);
u32  r4x16_dec_direct_budget(int nblk, const R4Opts *);
u32  r4x16_dec_mid_budget(int nblk, const R4Opts *);
u32  r4x16_enc_direct_budget(int nblk, const R4Opts *);
void r4x16_launch_stripe(const u8 *, u8 *, u32, u32, int, hipStream_t);
}

struct TimedLaunch { hipEvent_t a, b; };
struct HostPipe;
void r4x16_pipe_destroy(HostPipe *);
int r4x16_ensure_stage(rans4x16_hip_ctx *c, size_t bytes);
void r4x16_trim(rans4x16_hip_ctx *c, size_t keep);
// stream ordering of a context's arenas (workspace, stripe arena) between calls on different streams: r4x16_api.hip
int r4x16_ws_order_begin(rans4x16_hip_ctx *c, hipStream_t s);
int r4x16_ws_order_end(rans4x16_hip_ctx *c, hipStream_t s);
int r4x16_stripe_compress_dev(rans4x16_hip_ctx *c, int n, const BatchArgs &a, int order, uint32_t max_in_size, hipStream_t s);
int r4x16_stripe_uncompress_dev(rans4x16_hip_ctx *c, int n, const BatchArgs &a, uint32_t max_in_size, uint32_t max_out_cap,
                                uint32_t max_stripe_out, hipStream_t s);
int r4x16_run_host_batch(rans4x16_hip_ctx *c, int n, bool decode,
                          const unsigned char *const *in, const unsigned int *in_size,
                          unsigned char *const *out, unsigned int *out_size, const int *order, int *status);

struct rans4x16_hip_ctx {
    int device = 0;
    std::string err;
    R4Opts opts = {};                       // rans4x16_hip_set_option; defaults from the environment, read once per process
    // one growing device workspace
    u8 *ws = nullptr;
    size_t ws_bytes = 0;
    double *logtab = nullptr;
    u32 *rcptab = nullptr;
    // staging for the host-buffer entry points
    u8 *stage = nullptr;
    size_t stage_bytes = 0;
    // timing hook
    int timing = 0;
    std::vector<TimedLaunch> timed[2];
    size_t max_ws = (size_t)160 << 30;       // ceiling for one chunk of blocks (plan_chunk also looks at free memory)
    // X_STRIPE in the device-resident calls (r4x16_stripe.hip): the arena of the internal items, and how many planes a
    // device-resident decode batch reserves per block (0: stripe blocks report UNSUPPORTED there)
    u8 *xs = nullptr;
    size_t xs_bytes = 0;
    int dev_stripe_planes = 0;
    unsigned int dev_stripe_out = 0;        // largest uncompressed stripe block such a batch may hold
    bool in_stripe = false;
    // calls on different streams are ordered on the one workspace through this event
    hipEvent_t ws_done = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_busy = false;
    // side streams for the class launches of small batches (created at first use)
    R4Fork fork = {};
    bool fork_made = false;
    SchedHint hint[2] = {};    // [0] encode, [1] decode: the last batch's work per class (pinned; r4x16_sched.h)
    bool no_fork = false;                   // a lane of the host pipeline: the lanes are its concurrency (one priority each)
    // host-buffer batches: this context's own stream, and the lane contexts large batches are pipelined over
    hipStream_t stream = nullptr;
    struct HostPipe *pipe = nullptr;
};

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

