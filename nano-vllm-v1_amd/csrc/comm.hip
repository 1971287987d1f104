// Tensor-parallel exchange over xGMI without a collective library on the data path.
//
// Every rank owns one uncached (fine-grained) device region, exported to its peers through HIP IPC:
//     [ flags: kMaxBlocks x 8 x u32 | staging: max_bytes ]
// An all-reduce is ONE kernel per rank.  Block i of every rank works on the same element range i, so a
// per-block flag exchange between the ranks is all the synchronisation there is:
//     (stage range i of the input into the own staging area)  ->  barrier  ->  sum range i over all ranks'
//     staging areas in rank order (every rank gets identical bits)  ->  barrier (staging may be rewritten).
// Messages above kTwoShotBytes reduce-scatter in place and then gather, reading 2(w-1)/w of the message per
// rank instead of (w-1).  Flags are monotonically increasing epochs kept in device memory, so the kernels
// replay inside a hipGraph.  Every spin is bounded by a wall-clock limit; on expiry the rank records the
// fault (nvh_comm_error) and lets the grid drain instead of hanging the device.
//
// Replaces the NCCL all-reduce of RowParallelLinear.forward (reference layers/linear.py:407-409) and of
// VocabParallelEmbedding.forward (layers/embed_head.py:128-129).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int kMaxWorld = 8;
constexpr int kMaxBlocks = 128;
constexpr int kThreads = 512;
constexpr int64_t kFlagBytes = (int64_t)kMaxBlocks * kMaxWorld * sizeof(uint32_t);
constexpr int64_t kTwoShotBytes = 1 << 20;
constexpr uint64_t kSpinLimitTicks = 20ull * 100000000ull;   // 20 s of the 100 MHz wall clock

struct CommDev {                    // passed to kernels by value
    uint32_t* flags[kMaxWorld];     // flags[r]: rank r's flag block (own or IPC-mapped)
    bf16* stage[kMaxWorld];         // stage[r]: rank r's staging area
    uint32_t* epoch;                // [kMaxBlocks] own device memory: barriers passed so far per block
    uint32_t* fault;                // own device memory: [0] nonzero once a spin timed out (1 + silent rank),
                                    // [1..3] block, expected epoch, epoch seen of the first timeout
    uint64_t spin_limit;            // wall-clock ticks (100 MHz) a flag may stay behind
    int rank, world;
};

}  // namespace

struct nvh_comm {
    CommDev dev;
    void* region;                   // own uncached region
    void* peer_base[kMaxWorld];     // IPC-mapped peer regions (nullptr for self)
    void* local;                    // epoch + fault words
    int64_t max_bytes;
    bool opened;
};

namespace {

// Epoch e has been reached by every rank's block blockIdx.x once this returns.  `release`: this block wrote
// data the peers will read after the barrier.
__device__ __forceinline__ void comm_barrier(const CommDev& c, uint32_t e, bool release) {
    // Every thread that stored data for the peers fences at system scope.  When an EARLIER kernel of the stream
    // produced the data in place (a GEMM writing into the staging area), two things make it visible: the region is
    // allocated hipDeviceMallocUncached, so stores to it bypass the L2s and have reached memory when that kernel's
    // waves retire, and one lane still issues a system-scope fence here so the flag store below is ordered after
    // whatever this XCD's L2 might hold (no reliance on the scope of the fence between two graph nodes).
    if (release || threadIdx.x == 0) __threadfence_system();
    __syncthreads();
    const int t = threadIdx.x;
    if (t < c.world) {
        __hip_atomic_store(c.flags[t] + blockIdx.x * kMaxWorld + c.rank, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* mine = c.flags[c.rank] + blockIdx.x * kMaxWorld + t;
        const uint64_t t0 = wall_clock64();
        uint32_t seen;
        while ((int32_t)((seen = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) - e) < 0) {
            __builtin_amdgcn_s_sleep(2);
            // after the first timeout nothing waits any more: the step finishes (with garbage) and the host,
            // which checks the fault word, raises instead of sitting through one timeout per exchange
            if (wall_clock64() - t0 > c.spin_limit || __hip_atomic_load(c.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                if (atomicCAS(c.fault, 0u, 1u + (uint32_t)t) == 0u) {
                    c.fault[1] = blockIdx.x;
                    c.fault[2] = e;
                    c.fault[3] = seen;
                }
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: drop every line cached before the barrier
}

__device__ __forceinline__ bf16x8 sum_ranks(const CommDev& c, int64_t pack) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    bf16x8 v[kMaxWorld];
#pragma unroll
    for (int r = 0; r < kMaxWorld; ++r)
        if (r < c.world) v[r] = reinterpret_cast<const bf16x8*>(c.stage[r])[pack];
#pragma unroll
    for (int r = 0; r < kMaxWorld; ++r)
        if (r < c.world)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[r][j]);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[j]);
    return o;
}

// packs: 16-byte units of the message.  Block b owns packs [b*per, min((b+1)*per, packs)).
__global__ __launch_bounds__(kThreads) void allreduce_oneshot_kernel(CommDev c, const bf16* __restrict__ in, bf16* __restrict__ out,
                                                                     int64_t packs, int64_t per, int staged) {
    const uint32_t e0 = c.epoch[blockIdx.x];
    const int64_t lo = blockIdx.x * per, hi = min(lo + per, packs);
    if (staged)
        for (int64_t p = lo + threadIdx.x; p < hi; p += kThreads)
            reinterpret_cast<bf16x8*>(c.stage[c.rank])[p] = reinterpret_cast<const bf16x8*>(in)[p];
    comm_barrier(c, e0 + 1, staged);
    for (int64_t p = lo + threadIdx.x; p < hi; p += kThreads) reinterpret_cast<bf16x8*>(out)[p] = sum_ranks(c, p);
    comm_barrier(c, e0 + 2, false);
    if (threadIdx.x == 0) c.epoch[blockIdx.x] = e0 + 2;
}

// Slice s = packs [s*slice, min((s+1)*slice, packs)) is reduced by rank s.  Block b owns sub-range
// [b*per, (b+1)*per) of EVERY slice, on every rank, so the per-block barrier covers all the hand-offs.
__global__ __launch_bounds__(kThreads) void allreduce_twoshot_kernel(CommDev c, const bf16* __restrict__ in, bf16* __restrict__ out,
                                                                     int64_t packs, int64_t slice, int64_t per, int staged) {
    const uint32_t e0 = c.epoch[blockIdx.x];
    const int64_t sub_lo = blockIdx.x * per, sub_hi = min(sub_lo + per, slice);
    if (staged)
        for (int s = 0; s < c.world; ++s)
            for (int64_t q = sub_lo + threadIdx.x; q < sub_hi; q += kThreads) {
                const int64_t p = s * slice + q;
                if (p < packs) reinterpret_cast<bf16x8*>(c.stage[c.rank])[p] = reinterpret_cast<const bf16x8*>(in)[p];
            }
    comm_barrier(c, e0 + 1, staged);
    for (int64_t q = sub_lo + threadIdx.x; q < sub_hi; q += kThreads) {          // reduce own slice in place
        const int64_t p = c.rank * slice + q;
        if (p < packs) reinterpret_cast<bf16x8*>(c.stage[c.rank])[p] = sum_ranks(c, p);
    }
    comm_barrier(c, e0 + 2, true);
    for (int s = 0; s < c.world; ++s)
        for (int64_t q = sub_lo + threadIdx.x; q < sub_hi; q += kThreads) {
            const int64_t p = s * slice + q;
            if (p < packs) reinterpret_cast<bf16x8*>(out)[p] = reinterpret_cast<const bf16x8*>(c.stage[s])[p];
        }
    comm_barrier(c, e0 + 3, false);
    if (threadIdx.x == 0) c.epoch[blockIdx.x] = e0 + 3;
}

// Column-wise concatenation on `root`: out[row, r*cols + j] = rank r's in[row, j].  Same block-range ownership as
// the one-shot all-reduce; ranks other than root only stage and take part in the two barriers.
__global__ __launch_bounds__(kThreads) void gather_cols_kernel(CommDev c, const bf16* __restrict__ in, bf16* __restrict__ out,
                                                               int64_t cols8, int64_t packs, int64_t per, int staged, int root) {
    const uint32_t e0 = c.epoch[blockIdx.x];
    const int64_t lo = blockIdx.x * per, hi = min(lo + per, packs);
    if (staged)
        for (int64_t p = lo + threadIdx.x; p < hi; p += kThreads)
            reinterpret_cast<bf16x8*>(c.stage[c.rank])[p] = reinterpret_cast<const bf16x8*>(in)[p];
    comm_barrier(c, e0 + 1, staged);
    if (c.rank == root)
        for (int64_t p = lo + threadIdx.x; p < hi; p += kThreads) {
            const int64_t row = p / cols8, col = p - row * cols8;
            for (int r = 0; r < c.world; ++r)
                reinterpret_cast<bf16x8*>(out)[(row * c.world + r) * cols8 + col] = reinterpret_cast<const bf16x8*>(c.stage[r])[p];
        }
    comm_barrier(c, e0 + 2, false);
    if (threadIdx.x == 0) c.epoch[blockIdx.x] = e0 + 2;
}

// One-shot all-reduce fused with the residual add + RMSNorm that consumes it (the pattern after every row-parallel
// projection): block b owns rows [b*rpb, (b+1)*rpb) on every rank, one wave per row.  Per row: sum of the ranks'
// staged rows in rank order (fp32, rounded to bf16 exactly like the stand-alone all-reduce), + residual -> new
// residual, RMSNorm -> y.  Same bits as nvh_allreduce_bf16 followed by nvh_add_rmsnorm.
template <int MAXC>
__global__ __launch_bounds__(kThreads) void allreduce_add_rmsnorm_kernel(CommDev c, const bf16* __restrict__ in, bf16* __restrict__ residual,
                                                                         const bf16* __restrict__ w, bf16* __restrict__ y, int64_t rows,
                                                                         int dim, float eps, int64_t rpb, int staged) {
    const uint32_t e0 = c.epoch[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r_lo = blockIdx.x * rpb, r_hi = min(r_lo + rpb, rows);
    const int nchunk = dim >> 3;
    if (staged) {
        const int64_t p_lo = r_lo * nchunk, p_hi = r_hi * nchunk;
        for (int64_t p = p_lo + threadIdx.x; p < p_hi; p += kThreads)
            reinterpret_cast<bf16x8*>(c.stage[c.rank])[p] = reinterpret_cast<const bf16x8*>(in)[p];
    }
    comm_barrier(c, e0 + 1, staged);
    for (int64_t r = r_lo + wave; r < r_hi; r += kThreads / 64) {
        float v[MAXC][8];
        // The sum of squares follows the stand-alone norm kernels bit for bit: rows of up to 1024 elements are one
        // wave there too (one partial per lane); wider rows are four waves, thread t owning chunks t, t + 256, ..,
        // reduced per wave and then (w0 + w1) + (w2 + w3) - here chunk lane + 64 i is that kernel's wave i mod 4.
        const bool wide = dim > 1024;
        float ssw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                const bf16x8 s8 = sum_ranks(c, r * nchunk + ch);
                const bf16x8 res = *reinterpret_cast<const bf16x8*>(residual + r * (int64_t)dim + ch * 8);
                bf16x8 nr;
                float part = wide ? ssw[i & 3] : ssw[0];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = bf2f(s8[j]) + bf2f(res[j]);
                    nr[j] = f2bf(x);
                    v[i][j] = x;
                    part += x * x;
                }
                if (wide) ssw[i & 3] = part;
                else ssw[0] = part;
                *reinterpret_cast<bf16x8*>(residual + r * (int64_t)dim + ch * 8) = nr;
            }
        }
        float ss = wave_sum(ssw[0]);
        if (wide) {
            const float s1 = wave_sum(ssw[1]), s2 = wave_sum(ssw[2]), s3 = wave_sum(ssw[3]);
            ss = (ss + s1) + (s2 + s3);
        }
        const float inv = 1.0f / sqrtf(ss / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + ch * 8);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(v[i][j] * inv)) * bf2f(wv[j]));
                *reinterpret_cast<bf16x8*>(y + r * (int64_t)dim + ch * 8) = o;
            }
        }
    }
    comm_barrier(c, e0 + 2, false);
    if (threadIdx.x == 0) c.epoch[blockIdx.x] = e0 + 2;
}

}  // namespace

extern "C" int32_t nvh_comm_handle_bytes(void) { return (int32_t)sizeof(hipIpcMemHandle_t); }

extern "C" int32_t nvh_comm_create(int32_t rank, int32_t world, int64_t max_bytes, nvh_comm** out) {
    NVH_REQUIRE(out, NVH_EINVAL, "nvh_comm_create: null out");
    NVH_REQUIRE(world >= 2 && world <= kMaxWorld && rank >= 0 && rank < world, NVH_EINVAL, "nvh_comm_create: rank %d / world %d (2..%d)",
                rank, world, kMaxWorld);
    NVH_REQUIRE(max_bytes > 0 && max_bytes % 16 == 0, NVH_EINVAL, "nvh_comm_create: max_bytes must be a positive multiple of 16");
    nvh_comm* c = new nvh_comm();
    memset(c, 0, sizeof(*c));
    c->max_bytes = max_bytes;
    c->dev.rank = rank;
    c->dev.world = world;
    hipError_t e = hipExtMallocWithFlags(&c->region, kFlagBytes + max_bytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(c->region, 0, kFlagBytes);
    if (e == hipSuccess) e = hipMalloc(&c->local, (kMaxBlocks + 4) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->local, 0, (kMaxBlocks + 4) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        nvh_set_error("nvh_comm_create: %s", hipGetErrorString(e));
        if (c->region) (void)hipFree(c->region);
        if (c->local) (void)hipFree(c->local);
        delete c;
        return NVH_ELAUNCH;
    }
    c->dev.epoch = (uint32_t*)c->local;
    c->dev.fault = c->dev.epoch + kMaxBlocks;
    const char* ms = getenv("NVH_COMM_TIMEOUT_MS");            // default 20 s; tests and debugging shorten it
    c->dev.spin_limit = ms && atoll(ms) > 0 ? (uint64_t)atoll(ms) * 100000ull : kSpinLimitTicks;
    c->dev.flags[rank] = (uint32_t*)c->region;
    c->dev.stage[rank] = (bf16*)((char*)c->region + kFlagBytes);
    *out = c;
    return NVH_OK;
}

extern "C" int32_t nvh_comm_get_handle(nvh_comm* c, void* handle_out) {
    NVH_REQUIRE(c && handle_out, NVH_EINVAL, "nvh_comm_get_handle: null pointer");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, c->region);
    NVH_REQUIRE(e == hipSuccess, NVH_ELAUNCH, "nvh_comm_get_handle: %s", hipGetErrorString(e));
    memcpy(handle_out, &h, sizeof(h));
    return NVH_OK;
}

extern "C" int32_t nvh_comm_open_peers(nvh_comm* c, const void* handles) {
    NVH_REQUIRE(c && handles, NVH_EINVAL, "nvh_comm_open_peers: null pointer");
    NVH_REQUIRE(!c->opened, NVH_EINVAL, "nvh_comm_open_peers: already opened");
    for (int r = 0; r < c->dev.world; ++r) {
        if (r == c->dev.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
        void* base = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
        NVH_REQUIRE(e == hipSuccess, NVH_ELAUNCH, "nvh_comm_open_peers: rank %d cannot map rank %d: %s", c->dev.rank, r, hipGetErrorString(e));
        c->peer_base[r] = base;
        c->dev.flags[r] = (uint32_t*)base;
        c->dev.stage[r] = (bf16*)((char*)base + kFlagBytes);
    }
    c->opened = true;
    return NVH_OK;
}

extern "C" void* nvh_comm_buffer(nvh_comm* c) { return c ? (void*)c->dev.stage[c->dev.rank] : nullptr; }
extern "C" int64_t nvh_comm_max_bytes(nvh_comm* c) { return c ? c->max_bytes : 0; }

extern "C" int32_t nvh_allreduce_bf16(nvh_comm* c, const void* in, void* out, int64_t count, nvh_stream_t stream) {
    NVH_REQUIRE(c && in && out, NVH_EINVAL, "nvh_allreduce_bf16: null pointer");
    NVH_REQUIRE(c->opened, NVH_EINVAL, "nvh_allreduce_bf16: peers not mapped (nvh_comm_open_peers)");
    NVH_REQUIRE(count >= 0 && count % 8 == 0 && count * 2 <= c->max_bytes, NVH_EINVAL,
                "nvh_allreduce_bf16: count %ld must be a multiple of 8 and fit %ld staging bytes", (long)count, (long)c->max_bytes);
    NVH_REQUIRE(nvh_aligned16(in) && nvh_aligned16(out), NVH_EINVAL, "nvh_allreduce_bf16: 16-byte alignment required");
    const char* own = (const char*)c->dev.stage[c->dev.rank];
    const bool in_place = (const char*)in == own;
    NVH_REQUIRE(in_place || (const char*)in + count * 2 <= own || (const char*)in >= own + c->max_bytes, NVH_EINVAL,
                "nvh_allreduce_bf16: input overlaps the staging area without starting at it");
    NVH_REQUIRE((const char*)out + count * 2 <= own || (const char*)out >= own + c->max_bytes, NVH_EINVAL,
                "nvh_allreduce_bf16: output must lie outside the staging area");
    if (count == 0) return NVH_OK;
    const int64_t packs = count / 8;
    hipStream_t st = (hipStream_t)stream;
    if (count * 2 <= kTwoShotBytes) {
        const int nb = (int)std::min<int64_t>(kMaxBlocks / 2, nvh_cdiv(packs, kThreads));
        const int64_t per = nvh_cdiv(packs, nb);
        allreduce_oneshot_kernel<<<nb, kThreads, 0, st>>>(c->dev, (const bf16*)in, (bf16*)out, packs, per, in_place ? 0 : 1);
    } else {
        const int64_t slice = nvh_cdiv(packs, c->dev.world);
        const int nb = (int)std::min<int64_t>(kMaxBlocks, nvh_cdiv(slice, kThreads));
        const int64_t per = nvh_cdiv(slice, nb);
        allreduce_twoshot_kernel<<<nb, kThreads, 0, st>>>(c->dev, (const bf16*)in, (bf16*)out, packs, slice, per, in_place ? 0 : 1);
    }
    NVH_CHECK_LAUNCH("nvh_allreduce_bf16");
    return NVH_OK;
}

extern "C" int32_t nvh_allreduce_add_rmsnorm_bf16(nvh_comm* c, const void* in, void* residual, const void* weight, void* y,
                                                  int64_t rows, int64_t dim, float eps, nvh_stream_t stream) {
    NVH_REQUIRE(c && in && residual && weight && y, NVH_EINVAL, "nvh_allreduce_add_rmsnorm_bf16: null pointer");
    NVH_REQUIRE(c->opened, NVH_EINVAL, "nvh_allreduce_add_rmsnorm_bf16: peers not mapped (nvh_comm_open_peers)");
    NVH_REQUIRE(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= 8192 && rows * dim * 2 <= c->max_bytes, NVH_EINVAL,
                "nvh_allreduce_add_rmsnorm_bf16: [%ld, %ld] needs dim %% 8 == 0, dim <= 8192 and must fit %ld staging bytes", (long)rows,
                (long)dim, (long)c->max_bytes);
    NVH_REQUIRE(nvh_aligned16(in) && nvh_aligned16(residual) && nvh_aligned16(weight) && nvh_aligned16(y), NVH_EINVAL,
                "nvh_allreduce_add_rmsnorm_bf16: 16-byte alignment required");
    const char* own = (const char*)c->dev.stage[c->dev.rank];
    const bool in_place = (const char*)in == own;
    NVH_REQUIRE(in_place || (const char*)in + rows * dim * 2 <= own || (const char*)in >= own + c->max_bytes, NVH_EINVAL,
                "nvh_allreduce_add_rmsnorm_bf16: input overlaps the staging area without starting at it");
    if (rows == 0) return NVH_OK;
    const int nb = (int)std::min<int64_t>(kMaxBlocks, nvh_cdiv(rows, kThreads / 64));
    const int64_t rpb = nvh_cdiv(rows, nb);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_ARN(MAXC)                                                                                                   \
    allreduce_add_rmsnorm_kernel<MAXC><<<nb, kThreads, 0, st>>>(c->dev, (const bf16*)in, (bf16*)residual, (const bf16*)weight, \
                                                                (bf16*)y, rows, (int)dim, eps, rpb, in_place ? 0 : 1)
    if (dim <= 1024) LAUNCH_ARN(2);
    else if (dim <= 4096) LAUNCH_ARN(8);
    else LAUNCH_ARN(16);
#undef LAUNCH_ARN
    NVH_CHECK_LAUNCH("nvh_allreduce_add_rmsnorm_bf16");
    return NVH_OK;
}

extern "C" int32_t nvh_gather_cols_bf16(nvh_comm* c, const void* in, void* out, int64_t rows, int64_t cols, int32_t root,
                                        nvh_stream_t stream) {
    NVH_REQUIRE(c && in, NVH_EINVAL, "nvh_gather_cols_bf16: null pointer");
    NVH_REQUIRE(c->opened, NVH_EINVAL, "nvh_gather_cols_bf16: peers not mapped (nvh_comm_open_peers)");
    NVH_REQUIRE(root >= 0 && root < c->dev.world && (c->dev.rank != root || out), NVH_EINVAL, "nvh_gather_cols_bf16: bad root / null out on root");
    NVH_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && rows * cols * 2 <= c->max_bytes, NVH_EINVAL,
                "nvh_gather_cols_bf16: [%ld, %ld] needs cols %% 8 == 0 and must fit %ld staging bytes", (long)rows, (long)cols, (long)c->max_bytes);
    NVH_REQUIRE(nvh_aligned16(in) && nvh_aligned16(out), NVH_EINVAL, "nvh_gather_cols_bf16: 16-byte alignment required");
    const char* own = (const char*)c->dev.stage[c->dev.rank];
    const bool in_place = (const char*)in == own;
    NVH_REQUIRE(in_place || (const char*)in + rows * cols * 2 <= own || (const char*)in >= own + c->max_bytes, NVH_EINVAL,
                "nvh_gather_cols_bf16: input overlaps the staging area without starting at it");
    if (rows == 0) return NVH_OK;
    const int64_t packs = rows * cols / 8;
    const int nb = (int)std::min<int64_t>(kMaxBlocks, nvh_cdiv(packs, kThreads));
    gather_cols_kernel<<<nb, kThreads, 0, (hipStream_t)stream>>>(c->dev, (const bf16*)in, (bf16*)out, cols / 8, packs, nvh_cdiv(packs, nb),
                                                                 in_place ? 0 : 1, root);
    NVH_CHECK_LAUNCH("nvh_gather_cols_bf16");
    return NVH_OK;
}

// Synchronises the device; returns 0 when no exchange has timed out, else 1 + the first silent rank seen.
extern "C" int32_t nvh_comm_error(nvh_comm* c) {
    if (!c) return 0;
    uint32_t f[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpy(f, c->dev.fault, sizeof(f), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        nvh_set_error("nvh_comm_error: %s", hipGetErrorString(e));
        return NVH_ELAUNCH;
    }
    if (f[0])
        nvh_set_error("tensor-parallel exchange timed out: rank %d, block %u waited for rank %u to reach epoch %u (saw %u)",
                      c->dev.rank, f[1], f[0] - 1, f[2], f[3]);
    return (int32_t)f[0];
}

extern "C" int32_t nvh_comm_destroy(nvh_comm* c) {
    if (!c) return NVH_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->dev.world; ++r)
        if (c->peer_base[r]) (void)hipIpcCloseMemHandle(c->peer_base[r]);
    (void)hipFree(c->region);
    (void)hipFree(c->local);
    delete c;
    return NVH_OK;
}
