// Cost of staging 1-KiB pieces of a K/V tile into LDS, per wave: LDS-DMA (global_load_lds_dwordx4) against a register
// round trip (global_load_dwordx4 + ds_write_b128).  One workgroup per CU, WAVES waves, each stages PIECES pieces per
// iteration from one of 8 slices of a small buffer (L2 hits, as the workgroups of one kv head share K/V), then waits and hits a barrier.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/stage_cost.hip -o gpurun_out/stage_cost ; run: gpurun_out/stage_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

template <int MODE, int PIECES>
__global__ __launch_bounds__(512) void stage_kernel(const char* __restrict__ src, int iters, unsigned long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), waves = blockDim.x >> 6;
    const uint32_t lds_base = (uint32_t)(uintptr_t)((LDS_AS char*)smem);
    const char* base = src + ((size_t)(blockIdx.x & 7) * waves + wave) * (size_t)(PIECES * 1024) * 8;   // 8 distinct streams (as 8 kv heads): L2 hits
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const char* tile = base + (size_t)(it & 7) * (PIECES * 1024);
        const uint32_t slot = lds_base + (uint32_t)((it & 1) * waves + wave) * (PIECES * 1024);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"((uint32_t)(lane * 16)), "s"(tile + p * 1024), "s"(slot + p * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 r[PIECES];
#pragma unroll
            for (int p = 0; p < PIECES; ++p) r[p] = *reinterpret_cast<const u32x4*>(tile + p * 1024 + lane * 16);
#pragma unroll
            for (int p = 0; p < PIECES; ++p)
                *reinterpret_cast<LDS_AS u32x4*>((LDS_AS char*)smem + (slot - lds_base) + p * 1024 + lane * 16) = r[p];
        }
        __syncthreads();
        acc += *reinterpret_cast<LDS_AS float*>((LDS_AS char*)smem + (slot - lds_base) + lane * 4);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    sink[blockIdx.x * blockDim.x + tid] = acc;
}

template <int MODE, int PIECES>
void run(const char* name, int waves, const char* src, unsigned long long* dcyc, float* sink) {
    const int iters = 200;
    const int smem = 2 * waves * PIECES * 1024;
    (void)hipFuncSetAttribute((const void*)stage_kernel<MODE, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int rep = 0; rep < 2; ++rep) stage_kernel<MODE, PIECES><<<256, waves * 64, smem>>>(src, iters, dcyc, sink);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    stage_kernel<MODE, PIECES><<<256, waves * 64, smem>>>(src, iters, dcyc, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/WG %d pieces/wave %2d: %7.1f memtime ticks per iteration, %6.1f per piece | %.1f us per launch, %.0f ns per iteration, %.2f TB/s\n",
           name, waves, PIECES, (double)c / iters, (double)c / iters / PIECES, ms * 1e3, ms * 1e6 / iters, 256.0 * waves * PIECES * 1024 * iters / (ms * 1e-3) / 1e12);
}

int main() {
    char* src; unsigned long long* dcyc; float* sink;
    hipMalloc(&src, 256ull * 8 * 16 * 1024 * 8 + (1 << 20));
    hipMemset(src, 1, 256ull * 8 * 16 * 1024 * 8);
    hipMalloc(&dcyc, 8); hipMalloc(&sink, 256 * 512 * 4);
    for (int waves : {4, 8}) {
        run<0, 4>("LDS-DMA", waves, src, dcyc, sink);
        run<1, 4>("global load + ds_write", waves, src, dcyc, sink);
        run<0, 8>("LDS-DMA", waves, src, dcyc, sink);
        run<1, 8>("global load + ds_write", waves, src, dcyc, sink);
        if (waves == 4) {
            run<0, 16>("LDS-DMA", waves, src, dcyc, sink);
            run<1, 16>("global load + ds_write", waves, src, dcyc, sink);
        }
    }
    return 0;
}
