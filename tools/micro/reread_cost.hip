// What it would cost the O projection of a small decode batch to merge the attention's key ranges in its own prologue: every one
// of its workgroups (8 waves, 8 or 16 output columns over the whole K) needs ALL merged rows, so each re-reads every partial
// record (rows x 16 heads x ranges x 520 B) from L2.  Here: WGS workgroups of 8 waves each read the same BYTES of one buffer with
// 8-byte loads per lane, LOADS of them in flight per wave at a time, and reduce them; against the same launch reading one record.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/reread_cost.hip -o gpurun_out/reread_cost ; run: gpurun_out/reread_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int LOADS>
__global__ __launch_bounds__(512) void reread_kernel(const f32x2* __restrict__ src, int batches, unsigned long long* cycles, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x2 a = {0.f, 0.f};
    for (int b = 0; b < batches; ++b) {
        const f32x2* p = src + ((size_t)(b * 8 + wave) * LOADS) * 64 + lane;
        f32x2 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) v[i] = p[(size_t)i * 64];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) a += v[i];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a[0] + a[1];
}

template <int LOADS>
float time_us(int wgs, int batches, const f32x2* src, unsigned long long* dcyc, float* sink, unsigned long long* ticks) {
    const int n = 50;
    for (int r = 0; r < 3; ++r) reread_kernel<LOADS><<<wgs, 512>>>(src, batches, dcyc, sink);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int r = 0; r < n; ++r) reread_kernel<LOADS><<<wgs, 512>>>(src, batches, dcyc, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(ticks, dcyc, 8, hipMemcpyDeviceToHost);
    return ms * 1e3f / n;
}

int main() {
    f32x2* src; unsigned long long* dcyc; float* sink;
    hipMalloc(&src, 8 << 20); hipMemset(src, 0, 8 << 20);
    hipMalloc(&dcyc, 8); hipMalloc(&sink, 256 * 512 * 4);
    unsigned long long t;
    for (int wgs : {128, 64}) {
        const float base = time_us<1>(wgs, 1, src, dcyc, sink, &t);
        printf("%3d workgroups: one record per wave          %6.2f us per launch (%llu ticks in workgroup 0)\n", wgs, base, t);
        struct { const char* what; int loads, batches; } cases[] = {
            {"1 row x 16 heads x 16 ranges (133 KB)", 32, 1}, {"1 row x 16 heads x 32 ranges (266 KB)", 64, 1},
            {"8 rows x 16 heads x 8 ranges (532 KB)", 64, 2}, {"16 rows x 16 heads x 4 ranges (532 KB)", 32, 4}};
        for (auto& c : cases) {
            const float us = c.loads == 32 ? time_us<32>(wgs, c.batches, src, dcyc, sink, &t) : time_us<64>(wgs, c.batches, src, dcyc, sink, &t);
            printf("%3d workgroups: %-40s %6.2f us per launch (+%.2f us), %llu ticks in workgroup 0, %.1f TB/s out of L2\n", wgs, c.what, us, us - base, t,
                   (double)wgs * 8 * c.loads * c.batches * 512 / (us - base) / 1e6);
        }
    }
    return 0;
}
