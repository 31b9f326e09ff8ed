// scan.hip -- device-wide exclusive prefix sums (u32 / u64), three launches:
// per-block reduce -> single-workgroup scan of block sums -> per-block scan + add.
// No inter-workgroup spinning: every launch is a plain data-parallel kernel.
//
// A block covers 4096 elements as 4 rows of 1024; a thread owns 4 consecutive elements of every row,
// so every load/store instruction is a fully coalesced 16 B (u32) or 32 B (u64) per lane.
#include "kiss_internal.hpp"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ROWS = 4;
constexpr int SCAN_VEC = 4;
constexpr int SCAN_ROW_ELEMS = SCAN_THREADS * SCAN_VEC;   // 1024
constexpr int SCAN_BLOCK = SCAN_ROWS * SCAN_ROW_ELEMS;    // 4096 elements per workgroup

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

template <typename T>
__device__ __forceinline__ void load_row(const T *__restrict__ in, uint64_t base, uint64_t count, T v[SCAN_VEC])
{
    if (base + SCAN_VEC <= count && ((reinterpret_cast<uintptr_t>(in + base) & (sizeof(T) * SCAN_VEC - 1)) == 0)) {
        if constexpr (sizeof(T) == 4) {
            uint4 q = *reinterpret_cast<const uint4 *>(in + base);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(in + base);
            ulonglong2 a = p[0], b = p[1];
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < SCAN_VEC; e++) v[e] = (base + e < count) ? in[base + e] : (T)0;
    }
}

template <typename T>
__device__ __forceinline__ void store_row(T *__restrict__ out, uint64_t base, uint64_t count, const T v[SCAN_VEC])
{
    if (base + SCAN_VEC <= count && ((reinterpret_cast<uintptr_t>(out + base) & (sizeof(T) * SCAN_VEC - 1)) == 0)) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<uint4 *>(out + base) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            ulonglong2 *p = reinterpret_cast<ulonglong2 *>(out + base);
            p[0] = make_ulonglong2(v[0], v[1]);
            p[1] = make_ulonglong2(v[2], v[3]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < SCAN_VEC; e++)
            if (base + e < count) out[base + e] = v[e];
    }
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const T *__restrict__ in, uint64_t count,
                                                             T *__restrict__ blocksums)
{
    __shared__ T lds[SCAN_THREADS / 64];
    const uint64_t bbase = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_VEC;
    T s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ROWS; r++) {
        T v[SCAN_VEC];
        load_row<T>(in, bbase + (uint64_t)r * SCAN_ROW_ELEMS, count, v);
        s += v[0] + v[1] + v[2] + v[3];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int w = 0; w < SCAN_THREADS / 64; w++) t += lds[w];
        blocksums[blockIdx.x] = t;
    }
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
    const int wave = threadIdx.x >> 6;
    T inc = wave_inclusive_scan<T>(s);
    if (lane_id() == 63) lds[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T acc = 0;
        for (int w = 0; w < 1024 / 64; w++) {
            T t = lds[w];
            lds[w] = acc;
            acc += t;
        }
    }
    __syncthreads();
    T run = inc - s + lds[wave];
    for (uint64_t i = beg; i < end; i++) {
        T t = data[i];
        data[i] = run;
        run += t;
    }
}

// ONE_BLOCK: the whole input fits one block (count <= SCAN_BLOCK): no block sums, a single launch
template <typename T, bool ONE_BLOCK>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const T *in, T *out, uint64_t count,
                                                            const T *__restrict__ blocksums)
{
    __shared__ T wsum[SCAN_ROWS][SCAN_THREADS / 64];
    const int wave = threadIdx.x >> 6;
    const uint64_t bbase = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_VEC;
    T v[SCAN_ROWS][SCAN_VEC];
    T c[SCAN_ROWS], inc[SCAN_ROWS];
#pragma unroll
    for (int r = 0; r < SCAN_ROWS; r++) {
        load_row<T>(in, bbase + (uint64_t)r * SCAN_ROW_ELEMS, count, v[r]);
        c[r] = v[r][0] + v[r][1] + v[r][2] + v[r][3];
    }
#pragma unroll
    for (int r = 0; r < SCAN_ROWS; r++) {
        inc[r] = wave_inclusive_scan<T>(c[r]);
        if (lane_id() == 63) wsum[r][wave] = inc[r];
    }
    __syncthreads();
    T run = ONE_BLOCK ? (T)0 : blocksums[blockIdx.x];
#pragma unroll
    for (int r = 0; r < SCAN_ROWS; r++) {
        T pre = run;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; w++) {
            T t = wsum[r][w];
            if (w < wave) pre += t;
            run += t;
        }
        pre += inc[r] - c[r];
        T o[SCAN_VEC];
        o[0] = pre;
        o[1] = o[0] + v[r][0];
        o[2] = o[1] + v[r][1];
        o[3] = o[2] + v[r][2];
        store_row<T>(out, bbase + (uint64_t)r * SCAN_ROW_ELEMS, count, o);
    }
}

template <typename T>
int scan_impl(kiss_hip_ctx *ctx, const T *in, T *out, uint64_t count)
{
    if (count == 0) return KISS_HIP_OK;
    const uint64_t nb = div_up(count, SCAN_BLOCK);
    if (nb > ctx->scan_tmp_cap) return KINTERNAL();
    T *bs = reinterpret_cast<T *>(ctx->scan_tmp);
    KTimer t(ctx, KISS_HIP_K_SCAN, count);
    if (nb == 1) { // the per-pass control arrays of the drivers are mostly this small: one launch instead of three
        hipLaunchKernelGGL((k_scan_final<T, true>), dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, in, out, count, (const T *)nullptr);
        KCHECK(hipGetLastError());
        return KISS_HIP_OK;
    }
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, count, bs);
    hipLaunchKernelGGL(k_scan_single<T>, dim3(1), dim3(1024), 0, ctx->stream, bs, nb);
    hipLaunchKernelGGL((k_scan_final<T, false>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, ctx->stream, in, out, count, bs);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

} // namespace

// zero-fill as an ordinary kernel on the ctx stream (keeps strict kernel-after-kernel ordering; used instead of
// hipMemsetAsync between dependent kernels)
__global__ void k_zero_u32(uint32_t *p, uint64_t count)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = 0;
}
int kiss_zero_u32(kiss_hip_ctx *ctx, void *p, uint64_t count_u32)
{
    if (!count_u32) return KISS_HIP_OK;
    hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)div_up(count_u32, 256)), dim3(256), 0, ctx->stream, (uint32_t *)p,
                       count_u32);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

// the same for any 32-bit value: exactly count_u32 words, four per thread where p is 16-byte aligned
__global__ void k_fill_u32(uint32_t *p, uint32_t v, uint64_t count, int vec)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const uint64_t full = count >> 2;
        if (i < full) reinterpret_cast<uint4 *>(p)[i] = make_uint4(v, v, v, v);
        else if (i - full < (count & 3)) p[4 * full + (i - full)] = v;
    } else if (i < count) {
        p[i] = v;
    }
}
int kiss_fill_u32(kiss_hip_ctx *ctx, void *p, uint32_t value, uint64_t count_u32)
{
    if (!count_u32) return KISS_HIP_OK;
    const int vec = ((uintptr_t)p & 15) == 0;
    const uint64_t threads = vec ? (count_u32 >> 2) + (count_u32 & 3) : count_u32;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)div_up(threads, 256)), dim3(256), 0, ctx->stream, (uint32_t *)p, value,
                       count_u32, vec);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

int kiss_scan_u32(kiss_hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t count)
{
    return scan_impl<uint32_t>(ctx, in, out, count);
}
int kiss_scan_u64(kiss_hip_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t count)
{
    return scan_impl<uint64_t>(ctx, in, out, count);
}
