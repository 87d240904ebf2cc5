// Write bandwidth of the decoder's output pattern on gfx950: every lane of a wave owns 2 KiB of a 128-KiB span and
// the wave writes ROW bytes of each lane's 2 KiB per step (the staging-row flush of k_decode), against the same
// bytes written as one contiguous stream.
//   hipcc --offload-arch=gfx950 -O2 tests/micro/store_pattern.hip -o build/store_pattern && build/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int ROW, bool NT>
__global__ void __launch_bounds__(256) k_rows(u32x4 *out, int waves_per_cu_hint)
{
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char *base = reinterpret_cast<char *>(out) + wave * (64u * 2048u);
    constexpr unsigned LPR = ROW / 16, RPI = 64 / LPR;            // lanes per row, rows per store instruction
    const u32x4 v = {lane, 1u, 2u, 3u};
    for (unsigned step = 0; step < 2048u / ROW; step++) {
#pragma unroll
        for (unsigned k = 0; k < LPR; k++) {
            const unsigned row = k * RPI + lane / LPR, chunk = (lane % LPR) * 16u;
            u32x4 *p = reinterpret_cast<u32x4 *>(base + row * 2048u + step * ROW + chunk);
            if (NT) __builtin_nontemporal_store(v, p);
            else *p = v;
        }
        // (what a block decode takes between two flushes, so that the stores of the waves interleave as they do there)
        __builtin_amdgcn_s_sleep(32);
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) k_stream(u32x4 *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32x4 v = {(unsigned)i, 1u, 2u, 3u};
    const size_t n = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < 16; r++) {
        if (NT) __builtin_nontemporal_store(v, out + i + (size_t)r * n);
        else out[i + (size_t)r * n] = v;
    }
}

template <class F>
static float timed(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;
    u32x4 *d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    const unsigned waves = (unsigned)(bytes / (64 * 2048)), grid = waves / 4;
#define RUN(ROW, NT)                                                                                        \
    {                                                                                                       \
        const float ms = timed([&] { hipLaunchKernelGGL((k_rows<ROW, NT>), dim3(grid), dim3(256), 0, 0, d, 0); }); \
        printf("rows of %3d bytes, 2 KiB apart%s: %6.3f ms  %7.1f GB/s\n", ROW, NT ? " (nt)" : "     ", ms, bytes / ms / 1e6); \
    }
    RUN(32, false) RUN(32, true) RUN(64, false) RUN(64, true) RUN(128, false) RUN(128, true) RUN(256, false) RUN(256, true)
    {
        const unsigned g = (unsigned)(bytes / 16 / 16 / 256);
        float ms = timed([&] { hipLaunchKernelGGL((k_stream<false>), dim3(g), dim3(256), 0, 0, d); });
        printf("contiguous stream             : %6.3f ms  %7.1f GB/s\n", ms, bytes / ms / 1e6);
        ms = timed([&] { hipLaunchKernelGGL((k_stream<true>), dim3(g), dim3(256), 0, 0, d); });
        printf("contiguous stream (nt)        : %6.3f ms  %7.1f GB/s\n", ms, bytes / ms / 1e6);
    }
    return 0;
}
