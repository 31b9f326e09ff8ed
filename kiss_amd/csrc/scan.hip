// scan.hip -- device-wide exclusive prefix sums (u32 / u64), three launches:
// per-block reduce -> single-workgroup scan of block sums -> per-block scan + add.
// No inter-workgroup spinning: every launch is a plain data-parallel kernel.
#include "kiss_internal.hpp"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ELEMS = 16;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ELEMS; // 4096 elements per workgroup

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a workgroup of NT threads; returns exclusive prefix,
// *total receives the workgroup total.  lds must hold NT/64 + 1 entries.
template <typename T, int NT>
__device__ __forceinline__ T block_exclusive_scan(T v, T *lds, T *total)
{
    const int wave = threadIdx.x >> 6;
    T inc = wave_inclusive_scan<T>(v);
    if (lane_id() == 63) lds[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T acc = 0;
        for (int w = 0; w < NT / 64; w++) {
            T t = lds[w];
            lds[w] = acc;
            acc += t;
        }
        lds[NT / 64] = acc;
    }
    __syncthreads();
    T res = inc - v + lds[wave];
    *total = lds[NT / 64];
    __syncthreads();
    return res;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const T *__restrict__ in, uint64_t count,
                                                             T *__restrict__ blocksums)
{
    __shared__ T lds[SCAN_THREADS / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ELEMS;
    T s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ELEMS; e++) {
        uint64_t i = base + e;
        if (i < count) s += in[i];
    }
    T total;
    (void)block_exclusive_scan<T, SCAN_THREADS>(s, lds, &total);
    if (threadIdx.x == 0) blocksums[blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(1024) void k_scan_single(T *__restrict__ data, uint64_t count)
{
    __shared__ T lds[1024 / 64 + 1];
    const uint64_t chunk = (count + 1023) / 1024;
    const uint64_t beg = (uint64_t)threadIdx.x * chunk;
    const uint64_t end = beg + chunk < count ? beg + chunk : count;
    T s = 0;
    for (uint64_t i = beg; i < end; i++) s += data[i];
    T total;
    T run = block_exclusive_scan<T, 1024>(s, lds, &total);
    for (uint64_t i = beg; i < end; i++) {
        T t = data[i];
        data[i] = run;
        run += t;
    }
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const T *in, T *out, uint64_t count,
                                                            const T *__restrict__ blocksums)
{
    __shared__ T lds[SCAN_THREADS / 64 + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ELEMS;
    T v[SCAN_ELEMS];
    T s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ELEMS; e++) {
        uint64_t i = base + e;
        v[e] = (i < count) ? in[i] : (T)0;
        s += v[e];
    }
    T total;
    T run = block_exclusive_scan<T, SCAN_THREADS>(s, lds, &total) + blocksums[blockIdx.x];
#pragma unroll
    for (int e = 0; e < SCAN_ELEMS; e++) {
        uint64_t i = base + e;
        if (i < count) out[i] = run;
        run += v[e];
    }
}

template <typename T>
int scan_impl(kiss_hip_ctx *ctx, const T *in, T *out, uint64_t count)
{
    if (count == 0) return KISS_HIP_OK;
    const uint64_t nb = div_up(count, SCAN_BLOCK);
    if (nb > ctx->scan_tmp_cap) return KISS_HIP_E_INTERNAL;
    T *bs = reinterpret_cast<T *>(ctx->scan_tmp);
    KTimer t(ctx, KISS_HIP_K_SCAN, count);
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, count, bs);
    hipLaunchKernelGGL(k_scan_single<T>, dim3(1), dim3(1024), 0, ctx->stream, bs, nb);
    hipLaunchKernelGGL(k_scan_final<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out, count, bs);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

} // namespace

int kiss_scan_u32(kiss_hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t count)
{
    return scan_impl<uint32_t>(ctx, in, out, count);
}
int kiss_scan_u64(kiss_hip_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t count)
{
    return scan_impl<uint64_t>(ctx, in, out, count);
}
