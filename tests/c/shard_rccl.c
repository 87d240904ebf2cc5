/* tests/c/shard_rccl.c -- the C recipe of INTEGRATION.md section 5 (one stream over several devices: two
 * ncclAllGather calls around plan / emit / stitch), compiled against librccl and libaec.so.0 and run with a
 * communicator of ONE rank: the stitched stream must be the stream aec_buffer_encode produces for the same
 * input.  TEST INFRASTRUCTURE (built and run by tests/test_gpu_shard.py on the GPU box). */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aec_gpu.h"
#include "libaec.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)

int main(void)
{
    const size_t n_bytes = (size_t)128 * 16 * 2 * 333 + 64;        /* 333 RSIs and a short one */
    aec_gpu_params p = {16, 16, 128, AEC_DATA_PREPROCESS};
    uint16_t *h_in = (uint16_t *)malloc(n_bytes);
    uint32_t x = 32768, s = 12345;
    for (size_t i = 0; i < n_bytes / 2; i++) {
        s = s * 1664525u + 1013904223u;
        x += (s >> 29) - 3;
        if ((i & 8191) < 600) x -= (s >> 29) - 3;                  /* constant stretches: zero blocks */
        h_in[i] = (uint16_t)x;
    }
    /* reference result through the host ABI of the same library */
    size_t cap = aec_gpu_encode_bound(&p, n_bytes);
    unsigned char *want = (unsigned char *)malloc(cap);
    struct aec_stream strm;
    memset(&strm, 0, sizeof strm);
    strm.bits_per_sample = 16; strm.block_size = 16; strm.rsi = 128; strm.flags = AEC_DATA_PREPROCESS;
    strm.next_in = (const unsigned char *)h_in; strm.avail_in = n_bytes;
    strm.next_out = want; strm.avail_out = cap;
    CHECK(aec_buffer_encode(&strm) == AEC_OK);
    const size_t want_len = strm.total_out;

    ncclUniqueId id;
    ncclComm_t comm;
    hipStream_t stream;
    CHECK(hipSetDevice(0) == hipSuccess);
    CHECK(ncclGetUniqueId(&id) == ncclSuccess);
    CHECK(ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);
    CHECK(hipStreamCreate(&stream) == hipSuccess);
    const unsigned world = 1, rank = 0;

    aec_gpu_ctx *ctx;
    CHECK(aec_gpu_create(&ctx) == AEC_OK);
    void *d_in, *d_out, *d_gathered, *d_stream;
    aec_gpu_enc_result *d_plan, *d_plans;
    uint64_t *d_total_bytes;
    const size_t slot_bytes = (cap + 15) & ~(size_t)15, stream_cap = world * slot_bytes + 64;
    CHECK(hipMalloc(&d_in, n_bytes + 16) == hipSuccess && hipMalloc(&d_out, slot_bytes + 16) == hipSuccess);
    CHECK(hipMalloc(&d_gathered, world * slot_bytes + 16) == hipSuccess && hipMalloc(&d_stream, stream_cap) == hipSuccess);
    CHECK(hipMalloc((void **)&d_plan, 24) == hipSuccess && hipMalloc((void **)&d_plans, world * 24) == hipSuccess);
    CHECK(hipMalloc((void **)&d_total_bytes, 8) == hipSuccess);
    CHECK(hipMemcpy(d_in, h_in, n_bytes, hipMemcpyHostToDevice) == hipSuccess);

    /* ---- INTEGRATION.md section 5, verbatim ---- */
    CHECK(aec_gpu_encode_plan_async(ctx, &p, d_in, n_bytes, d_plan /* 24 B */, stream) == AEC_OK);
    CHECK(ncclAllGather(d_plan, d_plans /* world * 24 B */, 24, ncclUint8, comm, stream) == ncclSuccess);
    CHECK(aec_gpu_encode_emit_planned_async(ctx, &p, d_in, n_bytes, d_out, slot_bytes, d_plans, rank, NULL, d_plan, stream) == AEC_OK);
    CHECK(ncclAllGather(d_out, d_gathered /* world * slot_bytes + 16 */, slot_bytes, ncclUint8, comm, stream) == ncclSuccess);
    CHECK(aec_gpu_stitch_async(d_gathered, slot_bytes, d_plans, world, d_stream, stream_cap, d_total_bytes, stream) == AEC_OK);
    CHECK(hipStreamSynchronize(stream) == hipSuccess);

    uint64_t total = 0;
    CHECK(hipMemcpy(&total, d_total_bytes, 8, hipMemcpyDeviceToHost) == hipSuccess);
    unsigned char *got = (unsigned char *)malloc(total + 1);
    CHECK(hipMemcpy(got, d_stream, total, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(total == want_len);
    CHECK(memcmp(got, want, want_len) == 0);
    printf("shard_rccl ok: %zu bytes in, %llu bytes out over RCCL (world 1), identical to aec_buffer_encode\n", n_bytes,
           (unsigned long long)total);
    ncclCommDestroy(comm);
    aec_gpu_destroy(ctx);
    return 0;
}
