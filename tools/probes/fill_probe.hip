// usage: fill_probe [workgroups=256] [regions=workgroups]
// L2 -> LDS fill rate per CU: LDS-DMA (global_load_lds) against loads into registers followed by ds_write, for a ring of
// STAGES tiles per workgroup (256 threads, one workgroup per CU).  Build: hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TILE_KB per stage; each wave moves TILE_KB/4 KB per tile = PW instructions of 1 KB
template <int TILE_KB, int STAGES, bool DMA>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, int64_t wg_stride, int tiles, float* sink, int regions) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PW = TILE_KB / 4;
    constexpr int AHEAD = STAGES - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (int64_t)(blockIdx.x % regions) * wg_stride + wave * (PW * 1024) + lane * 16;
    float acc = 0.f;
    if constexpr (DMA) {
        auto stage = [&](int t) {
            char* dst = smem + (t % STAGES) * (TILE_KB * 1024) + wave * (PW * 1024);
#pragma unroll
            for (int i = 0; i < PW; ++i)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(base + (int64_t)t * (TILE_KB * 1024) + i * 1024), (LDS_AS void*)(dst + i * 1024), 16, 0, 0);
        };
        for (int t = 0; t < AHEAD && t < tiles; ++t) stage(t);
        for (int t = 0; t < tiles; ++t) {
            if (t + AHEAD < tiles) { stage(t + AHEAD); wait_vmcnt<(PW * AHEAD < 64 ? PW * AHEAD : 63)>(); }
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            acc += *reinterpret_cast<const float*>(smem + (t % STAGES) * (TILE_KB * 1024) + tid * 4);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // registers: AHEAD tiles of PW x 16 B per lane in flight, written to LDS when they land
        f32x4 r[AHEAD > 0 ? AHEAD : 1][PW];
        auto load = [&](int t, int slot) {
#pragma unroll
            for (int i = 0; i < PW; ++i) r[slot][i] = *reinterpret_cast<const f32x4*>(base + (int64_t)t * (TILE_KB * 1024) + i * 1024);
        };
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) if (t < tiles) load(t, t);
        for (int t0 = 0; t0 < tiles; t0 += AHEAD) {
#pragma unroll
            for (int s = 0; s < AHEAD; ++s) {
                const int t = t0 + s;
                if (t >= tiles) break;
                char* dst = smem + (t % STAGES) * (TILE_KB * 1024) + wave * (PW * 1024) + lane * 16;
#pragma unroll
                for (int i = 0; i < PW; ++i) *reinterpret_cast<f32x4*>(dst + i * 1024) = r[s][i];
                if (t + AHEAD < tiles) load(t + AHEAD, s);
                __builtin_amdgcn_s_barrier();
                acc += *reinterpret_cast<const float*>(smem + (t % STAGES) * (TILE_KB * 1024) + tid * 4);
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int TILE_KB, int STAGES, bool DMA>
void run(const char* src, int64_t bytes, float* sink, int wgs, int regions) {
    auto k = fill_kernel<TILE_KB, STAGES, DMA>;
    const int smem = TILE_KB * 1024 * STAGES;
    if (smem > 160 * 1024) return;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int tiles = 64;                                   // 64 tiles per workgroup
    const int64_t wg_stride = (int64_t)tiles * TILE_KB * 1024;
    if (wg_stride * regions > bytes) { printf("buffer too small\n"); return; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<wgs, 256, smem>>>(src, wg_stride, tiles, sink, regions);
    hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<<<wgs, 256, smem>>>(src, wg_stride, tiles, sink, regions);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    const double gb = (double)wgs * tiles * TILE_KB * 1024 / 1e9;
    printf("%s tile %2d KB x %d stages (%3d KB LDS), %d workgroups over %d regions: %7.1f us  %6.1f GB/s per workgroup  %6.2f TB/s chip\n", DMA ? "dma " : "vgpr", TILE_KB, STAGES,
           smem / 1024, wgs, regions, us, gb / (us * 1e-6) / wgs, gb / (us * 1e-6) / 1e3);
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int regions = argc > 2 ? atoi(argv[2]) : wgs;   // distinct source regions (workgroup i reads region i % regions)
    const int64_t bytes = (int64_t)wgs * 64 * 32 * 1024;    // up to 32 KB tiles: 512 MB at 256 workgroups (fits the Infinity Cache only partly)
    char* src; float* sink;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, bytes);
    run<16, 2, true>(src, bytes, sink, wgs, regions);  run<16, 4, true>(src, bytes, sink, wgs, regions);  run<16, 8, true>(src, bytes, sink, wgs, regions);
    run<32, 2, true>(src, bytes, sink, wgs, regions);  run<32, 3, true>(src, bytes, sink, wgs, regions);  run<32, 4, true>(src, bytes, sink, wgs, regions);
    run<16, 2, false>(src, bytes, sink, wgs, regions); run<16, 4, false>(src, bytes, sink, wgs, regions); run<16, 8, false>(src, bytes, sink, wgs, regions);
    run<32, 2, false>(src, bytes, sink, wgs, regions); run<32, 3, false>(src, bytes, sink, wgs, regions); run<32, 4, false>(src, bytes, sink, wgs, regions);
    hipDeviceSynchronize();
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
