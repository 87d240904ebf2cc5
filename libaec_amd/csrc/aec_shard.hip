// aec_shard.hip -- one adaptive-entropy stream coded on several devices (mode ii).
//
// The coder's only state that crosses a block boundary is the bit position and the carried k
// (reference src/encode.c: state->bits / state->cds and state->k), so contiguous RSI-aligned shards
// are planned independently (aec_gpu_encode_plan_async: total bits and the k clamp of the shard),
// the plan records are all-gathered (24 bytes per rank, RCCL), and everything else happens on the
// device without a host round trip:
//   k_shard_carry   start bit = sum of the preceding shards' bits, k_in = composition of their clamps
//   k_stitch        the all-gathered byte slices (slice r starts at bit start_r % 8 of its slot) are
//                   compacted into one stream; the <= world-1 bytes two slices share are OR-ed
#include <hip/hip_runtime.h>

#include "aec_kernels.h"
#include "aec_lane.h"

namespace aec {

namespace {

__global__ void k_shard_carry(const EncResult *__restrict__ plans, uint32_t rank, ShardCarry *carry)
{
    uint64_t start = 0;
    uint32_t k = 0;                                  // reference encode.c:800: a stream starts with k = 0
    for (uint32_t r = 0; r < rank; r++) {
        start += plans[r].total_bits;
        k = kclamp_apply(KClamp{plans[r].k_lo, plans[r].k_hi}, k);
    }
    carry->start_bit = start;
    carry->k_in = k;
    carry->pad = 0;
}

constexpr uint32_t kMaxWorld = 64;

// one thread = one aligned 16-byte vector of the OUTPUT stream
__global__ void __launch_bounds__(256)
k_stitch(const uint8_t *__restrict__ gathered, uint64_t slot, const EncResult *__restrict__ plans, uint32_t world,
         uint8_t *__restrict__ out, uint64_t cap, uint64_t *total_bytes)
{
    __shared__ uint64_t s_lo[kMaxWorld], s_hi[kMaxWorld], s_lead[kMaxWorld];
    __shared__ uint64_t s_total;
    if (threadIdx.x == 0) {
        uint64_t start = 0;
        for (uint32_t r = 0; r < world; r++) {
            const uint64_t bits = plans[r].total_bits;
            s_lo[r] = start >> 3;                               // first byte slice r touches
            s_hi[r] = bits ? (start + bits + 7) >> 3 : s_lo[r]; // one past its last byte
            s_lead[r] = start >> 3;                             // its slot byte 0 is stream byte start / 8
            start += bits;
        }
        s_total = start ? (start + 7) >> 3 : 1;                 // an empty stream is one zero byte
        if (blockIdx.x == 0 && total_bytes) *total_bytes = s_total;
    }
    __syncthreads();
    const uint64_t total = s_total < cap ? s_total : cap;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v * 16 < total;
         v += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b0 = v * 16, b1 = b0 + 16;
        // slices that touch this vector: r_first = the last slice starting at or before b0
        uint32_t inside = world;                    // slice that covers the vector alone
        bool mixed = false;
        for (uint32_t r = 0; r < world; r++) {
            if (s_hi[r] <= b0 || s_lo[r] >= b1 || s_hi[r] == s_lo[r]) continue;
            if (s_lo[r] <= b0 && s_hi[r] >= b1 && inside == world && !mixed) inside = r;
            else mixed = true;
        }
        // (a neighbour that shares only the boundary byte makes the vector "mixed" as well)
        if (inside != world && !mixed) {
            const uint64_t src = (uint64_t)inside * slot + (b0 - s_lead[inside]);
            const uint32_t sh = (uint32_t)(src & 3u);
            const uint32_t *w = reinterpret_cast<const uint32_t *>(gathered + (src & ~3ull));
            const uint32_t a0 = w[0], a1 = w[1], a2 = w[2], a3 = w[3], a4 = sh ? w[4] : 0u;
            uint4 o;
            o.x = __builtin_amdgcn_alignbyte(a1, a0, sh);
            o.y = __builtin_amdgcn_alignbyte(a2, a1, sh);
            o.z = __builtin_amdgcn_alignbyte(a3, a2, sh);
            o.w = __builtin_amdgcn_alignbyte(a4, a3, sh);
            if (b1 <= total) {
                *reinterpret_cast<uint4 *>(out + b0) = o;
                continue;
            }
        }
        // boundary vector (or the ragged end): byte by byte, OR over the slices that hold the byte
        for (uint64_t b = b0; b < b1 && b < total; b++) {
            uint8_t x = 0;
            for (uint32_t r = 0; r < world; r++)
                if (b >= s_lo[r] && b < s_hi[r]) x |= gathered[(uint64_t)r * slot + (b - s_lead[r])];
            out[b] = x;
        }
    }
}

}  // namespace

void launch_shard_carry(const EncResult *d_plans, uint32_t rank, ShardCarry *d_carry, hipStream_t st)
{
    hipLaunchKernelGGL(k_shard_carry, dim3(1), dim3(1), 0, st, d_plans, rank, d_carry);
}

void launch_stitch(const uint8_t *d_gathered, size_t slot, const EncResult *d_plans, uint32_t world,
                   uint8_t *d_stream, size_t cap, uint64_t *d_total_bytes, hipStream_t st)
{
    // grid-stride over the output; sized for the most the slots can hold
    const uint64_t vecs = ((uint64_t)slot * world + 15) / 16;
    uint64_t blocks = (vecs + 255) / 256;
    if (blocks > 256u * 64u) blocks = 256u * 64u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_stitch, dim3((uint32_t)blocks), dim3(256), 0, st, d_gathered, (uint64_t)slot, d_plans, world,
                       d_stream, (uint64_t)cap, d_total_bytes);
}

}  // namespace aec
