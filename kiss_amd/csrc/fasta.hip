// fasta.hip -- FASTA / plain-text file -> base codes 0..3, parsed on the GPU.
//
// Semantics restated from the reference reader (include/utils/io.hpp:6-18 + biovoltron/file_io/fasta.hpp:117-151 +
// Codec::to_int, biovoltron/utility/istring.hpp:28-46, then `c % 4` in command/suffix_sort.hpp:33):
//   * the file is FASTA iff its first byte is '>', otherwise plain text;
//   * plain text: every byte except '\n' is a base;
//   * FASTA: a record starts at a line that begins with '>' (its header, dropped); the line right after a header is
//     always sequence -- even if it begins with '>' (the reference reads it with an unconditional getline) -- and a
//     record ends in front of the next line that begins with '>'.  So in a run of consecutive '>' lines the 1st,
//     3rd, 5th ... are headers and the others are sequence;
//   * code: A/a 0, C/c 1, G/g 2, T/t 3, every other byte 4 % 4 = 0 (this includes '\r' of CRLF files and blanks:
//     std::getline only removes '\n').
// The host side only moves raw bytes (pread into pinned buffers, async copies); nothing touches a base on the CPU.
//
// Device passes over the raw bytes (tiles of 4096):
//   K1 per-tile counts of '\n' and of lines that begin with '>'      -> scans
//   K2 list of '>' lines (their line numbers), header = even place inside its run of consecutive '>' lines
//   K3 per-tile count of kept bytes (not '\n', not inside a header line) -> scan
//   K4 codes written compacted
#include "kiss_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int FA_THREADS = 256;
constexpr int FA_PER = 16;
constexpr int FA_TILE = FA_THREADS * FA_PER; // 4096 bytes per workgroup

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// per tile: number of '\n', number of lines beginning with '>'
__global__ __launch_bounds__(FA_THREADS) void k_fa_count(const uint8_t *__restrict__ raw, uint64_t bytes,
                                                        uint32_t *__restrict__ tile_nl, uint32_t *__restrict__ tile_gt)
{
    __shared__ uint32_t ws[2][FA_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * FA_PER;
    uint32_t nl = 0, gt = 0;
    uint8_t prev = (base > 0 && base <= bytes) ? raw[base - 1] : (uint8_t)'\n';
#pragma unroll
    for (int e = 0; e < FA_PER; e++) {
        const uint64_t i = base + e;
        if (i < bytes) {
            const uint8_t b = raw[i];
            nl += b == '\n';
            gt += (b == '>' && prev == '\n');
            prev = b;
        }
    }
    nl = wave_sum(nl);
    gt = wave_sum(gt);
    if (lane_id() == 0) {
        ws[0][threadIdx.x >> 6] = nl;
        ws[1][threadIdx.x >> 6] = gt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (int w = 0; w < FA_THREADS / 64; w++) {
            a += ws[0][w];
            b += ws[1][w];
        }
        tile_nl[blockIdx.x] = a;
        tile_gt[blockIdx.x] = b;
    }
}

// exclusive prefix over the threads of a workgroup of (a, b); returns this thread's offsets
__device__ __forceinline__ void block_excl2(uint32_t a, uint32_t b, uint32_t &ea, uint32_t &eb, uint32_t (*ws)[FA_THREADS / 64])
{
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t oa = __shfl_up(ia, d, 64), ob = __shfl_up(ib, d, 64);
        if ((int)lane_id() >= d) {
            ia += oa;
            ib += ob;
        }
    }
    const int wave = threadIdx.x >> 6;
    if (lane_id() == 63) {
        ws[0][wave] = ia;
        ws[1][wave] = ib;
    }
    __syncthreads();
    ea = ia - a;
    eb = ib - b;
    for (int w = 0; w < wave; w++) {
        ea += ws[0][w];
        eb += ws[1][w];
    }
    __syncthreads();
}

// line number of every line that begins with '>' (in file order)
__global__ __launch_bounds__(FA_THREADS) void k_fa_gt_lines(const uint8_t *__restrict__ raw, uint64_t bytes,
                                                           const uint32_t *__restrict__ nl_ex,
                                                           const uint32_t *__restrict__ gt_ex,
                                                           uint32_t *__restrict__ gt_line)
{
    __shared__ uint32_t ws[2][FA_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * FA_PER;
    uint8_t loc[FA_PER];
    uint32_t nl = 0, gt = 0;
    const uint8_t prev0 = (base > 0 && base <= bytes) ? raw[base - 1] : (uint8_t)'\n';
    uint8_t prev = prev0;
#pragma unroll
    for (int e = 0; e < FA_PER; e++) {
        const uint64_t i = base + e;
        loc[e] = i < bytes ? raw[i] : (uint8_t)0;
        if (i < bytes) {
            nl += loc[e] == '\n';
            gt += (loc[e] == '>' && prev == '\n');
            prev = loc[e];
        }
    }
    uint32_t enl, egt;
    block_excl2(nl, gt, enl, egt, ws);
    uint32_t line = nl_ex[blockIdx.x] + enl;
    uint32_t g = gt_ex[blockIdx.x] + egt;
    prev = prev0;
#pragma unroll
    for (int e = 0; e < FA_PER; e++) {
        const uint64_t i = base + e;
        if (i < bytes) {
            if (loc[e] == '>' && prev == '\n') gt_line[g++] = line;
            line += loc[e] == '\n';
            prev = loc[e];
        }
    }
}

// runs of consecutive line numbers in gt_line: start flag per entry
__global__ void k_fa_run_flags(const uint32_t *__restrict__ gt_line, uint64_t G, uint32_t *__restrict__ flag)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < G) flag[j] = (j == 0 || gt_line[j] != gt_line[j - 1] + 1u) ? 1u : 0u;
}
// run_ex = exclusive scan of flag; the start of run r is the entry j with flag[j] = 1 and run_ex[j] = r
__global__ void k_fa_run_starts(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ run_ex, uint64_t G,
                                uint32_t *__restrict__ run_start)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < G && flag[j]) run_start[run_ex[j]] = (uint32_t)j;
}
// is_header[j] = entry j sits at an even place of its run
__global__ void k_fa_headers(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ run_ex,
                             const uint32_t *__restrict__ run_start, uint64_t G, uint32_t *__restrict__ is_header)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= G) return;
    const uint32_t r = run_ex[j] + flag[j] - 1u; // run of entry j
    is_header[j] = (((uint32_t)j - run_start[r]) & 1u) ? 0u : 1u;
}

__device__ __forceinline__ bool line_is_header(const uint32_t *__restrict__ gt_line, const uint32_t *__restrict__ is_header,
                                               uint32_t G, uint32_t line)
{
    uint32_t lo = 0, hi = G; // first entry with gt_line >= line
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (gt_line[mid] < line) lo = mid + 1;
        else hi = mid;
    }
    return lo < G && gt_line[lo] == line && is_header[lo] != 0;
}

__device__ __forceinline__ uint8_t base_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0; // Codec::to_int = 4, then % 4
    }
}

// EMIT = false: per-tile count of kept bytes; EMIT = true: codes written at keep_ex[tile] + rank
template <bool EMIT>
__global__ __launch_bounds__(FA_THREADS) void k_fa_keep(const uint8_t *__restrict__ raw, uint64_t bytes,
                                                       const uint32_t *__restrict__ nl_ex,
                                                       const uint32_t *__restrict__ gt_line,
                                                       const uint32_t *__restrict__ is_header, uint32_t G,
                                                       uint32_t *__restrict__ tile_keep,
                                                       const uint32_t *__restrict__ keep_ex, uint8_t *__restrict__ out)
{
    __shared__ uint32_t ws[2][FA_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FA_TILE + (uint64_t)threadIdx.x * FA_PER;
    uint8_t loc[FA_PER];
    uint32_t nl = 0;
#pragma unroll
    for (int e = 0; e < FA_PER; e++) {
        const uint64_t i = base + e;
        loc[e] = i < bytes ? raw[i] : (uint8_t)'\n';
        nl += (i < bytes && loc[e] == '\n');
    }
    uint32_t enl, dummy;
    block_excl2(nl, 0u, enl, dummy, ws);
    uint32_t line = nl_ex[blockIdx.x] + enl;
    bool hdr = G ? line_is_header(gt_line, is_header, G, line) : false;
    uint32_t keepmask = 0, kept = 0;
#pragma unroll
    for (int e = 0; e < FA_PER; e++) {
        const uint64_t i = base + e;
        if (i < bytes) {
            if (loc[e] == '\n') {
                line++;
                hdr = G ? line_is_header(gt_line, is_header, G, line) : false;
            } else if (!hdr) {
                keepmask |= 1u << e;
                kept++;
            }
        }
    }
    uint32_t ek;
    block_excl2(kept, 0u, ek, dummy, ws);
    if (!EMIT) {
        if (threadIdx.x == FA_THREADS - 1) tile_keep[blockIdx.x] = ek + kept;
        return;
    }
    uint64_t o = (uint64_t)keep_ex[blockIdx.x] + ek;
#pragma unroll
    for (int e = 0; e < FA_PER; e++)
        if ((keepmask >> e) & 1u) out[o++] = base_code(loc[e]);
}

__global__ void k_fa_last(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ ex, uint64_t tiles,
                          uint32_t *__restrict__ total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = ex[tiles - 1] + cnt[tiles - 1];
}

int read_u32(kiss_hip_ctx *ctx, const uint32_t *d, uint32_t *out)
{
    KCHECK(hipMemcpyAsync(ctx->h_pinned, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    *out = ctx->h_pinned[0];
    return KISS_HIP_OK;
}

} // namespace

extern "C" {

int kiss_hip_ctx_parse_text_dev(kiss_hip_ctx *ctx, const uint8_t *d_raw, uint64_t bytes, uint8_t *d_S, uint64_t *n_out,
                                void *stream)
{
    if (!ctx || !n_out || (bytes && (!d_raw || !d_S))) return KISS_HIP_E_INVALID;
    if (bytes > 0xFFFFFFF0ull) return KISS_HIP_E_INVALID; // line numbers and offsets are 32-bit (n < 2^32 anyway)
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    KTRY(kiss_workspace_ready(ctx));
    *n_out = 0;
    if (bytes == 0) return KISS_HIP_OK;
    const uint64_t tiles = div_up(bytes, FA_TILE);
    if (tiles + 2 > ctx->m_cap) return KISS_HIP_E_INVALID; // ctx too small for this file (max_n >= bytes is enough)
    if (tiles + 2 > ctx->t_cap) KTRY(kiss_tied_reserve(ctx, tiles + 2));
restart:
    uint32_t *tile_nl = ctx->segA, *nl_ex = ctx->segB, *tile_gt = ctx->slotA, *gt_ex = ctx->slotB;
    uint32_t *tile_keep = ctx->bsegA, *keep_ex = ctx->bsegB;
    uint32_t *gt_line = ctx->posA, *flag = ctx->posB, *run_ex = ctx->bposA, *run_start = ctx->bposB, *is_header = ctx->bslot;
    uint32_t *d_tot = ctx->d_small + 12;
    uint8_t first = 0;
    KCHECK(hipMemcpyAsync(&first, d_raw, 1, hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    const bool fasta = first == '>';
    KTimer t(ctx, KISS_HIP_K_PACK, bytes);
    hipLaunchKernelGGL(k_fa_count, dim3((unsigned)tiles), dim3(FA_THREADS), 0, ctx->stream, d_raw, bytes, tile_nl, tile_gt);
    KCHECK(hipGetLastError());
    KTRY(kiss_scan_u32(ctx, tile_nl, nl_ex, tiles));
    uint32_t G = 0;
    if (fasta) {
        KTRY(kiss_scan_u32(ctx, tile_gt, gt_ex, tiles));
        hipLaunchKernelGGL(k_fa_last, dim3(1), dim3(64), 0, ctx->stream, tile_gt, gt_ex, tiles, d_tot);
        KTRY(read_u32(ctx, d_tot, &G));
        if ((uint64_t)G + 2 > ctx->m_cap) return KISS_HIP_E_UNSUPPORTED; // a third of the lines are 1-2 bytes long
        if ((uint64_t)G + 2 > ctx->t_cap) { // per-header scratch lives in the tied-segment arrays: regrow, start over
            KTRY(kiss_tied_reserve(ctx, (uint64_t)G + 2));
            goto restart;
        }
        hipLaunchKernelGGL(k_fa_gt_lines, dim3((unsigned)tiles), dim3(FA_THREADS), 0, ctx->stream, d_raw, bytes, nl_ex, gt_ex,
                           gt_line);
        const unsigned gb = (unsigned)div_up(G, 256);
        hipLaunchKernelGGL(k_fa_run_flags, dim3(gb), dim3(256), 0, ctx->stream, gt_line, (uint64_t)G, flag);
        KCHECK(hipGetLastError());
        KTRY(kiss_scan_u32(ctx, flag, run_ex, G));
        hipLaunchKernelGGL(k_fa_run_starts, dim3(gb), dim3(256), 0, ctx->stream, flag, run_ex, (uint64_t)G, run_start);
        hipLaunchKernelGGL(k_fa_headers, dim3(gb), dim3(256), 0, ctx->stream, flag, run_ex, run_start, (uint64_t)G, is_header);
        KCHECK(hipGetLastError());
    }
    hipLaunchKernelGGL((k_fa_keep<false>), dim3((unsigned)tiles), dim3(FA_THREADS), 0, ctx->stream, d_raw, bytes, nl_ex, gt_line,
                       is_header, G, tile_keep, keep_ex, d_S);
    KCHECK(hipGetLastError());
    KTRY(kiss_scan_u32(ctx, tile_keep, keep_ex, tiles));
    hipLaunchKernelGGL(k_fa_last, dim3(1), dim3(64), 0, ctx->stream, tile_keep, keep_ex, tiles, d_tot);
    hipLaunchKernelGGL((k_fa_keep<true>), dim3((unsigned)tiles), dim3(FA_THREADS), 0, ctx->stream, d_raw, bytes, nl_ex, gt_line,
                       is_header, G, tile_keep, keep_ex, d_S);
    KCHECK(hipGetLastError());
    uint32_t n32 = 0;
    KTRY(read_u32(ctx, d_tot, &n32));
    *n_out = n32;
    return KISS_HIP_OK;
}

int kiss_hip_file_size(const char *path, uint64_t *bytes)
{
    if (!path || !bytes) return KISS_HIP_E_INVALID;
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return KISS_HIP_E_IO;
    *bytes = (uint64_t)st.st_size;
    return KISS_HIP_OK;
}

int kiss_hip_ctx_load_text_file(kiss_hip_ctx *ctx, const char *path, uint8_t **d_S_out, uint64_t *n_out)
{
    if (!ctx || !path || !d_S_out || !n_out) return KISS_HIP_E_INVALID;
    *d_S_out = nullptr;
    *n_out = 0;
    uint64_t bytes = 0;
    int rc = kiss_hip_file_size(path, &bytes);
    if (rc) return rc;
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = ctx->own_stream;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return KISS_HIP_E_IO;
    uint8_t *d_raw = nullptr, *d_S = nullptr;
    // READERS threads, each with two pinned buffers and its own stream: the page-cache copy of one chunk overlaps the
    // upload of the previous one, and the threads overlap each other (one pread stream is ~2-3 GB/s)
    constexpr int READERS = 4;
    constexpr size_t CHUNK = 32ull << 20;
    struct Reader {
        void *pin[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        hipStream_t stream = nullptr;
    } rd[READERS];
    std::atomic<int> status{KISS_HIP_OK};
    rc = KISS_HIP_OK;
    do {
        if (hipMalloc((void **)&d_raw, bytes ? bytes : 1) != hipSuccess || hipMalloc((void **)&d_S, bytes ? bytes : 1) != hipSuccess) {
            rc = KISS_HIP_E_NOMEM;
            break;
        }
        const uint64_t chunks = div_up(bytes, CHUNK);
        const int readers = (int)(chunks < (uint64_t)READERS ? (chunks ? chunks : 1) : READERS);
        for (int t = 0; t < readers && rc == KISS_HIP_OK; t++) {
            if (hipStreamCreateWithFlags(&rd[t].stream, hipStreamNonBlocking) != hipSuccess) rc = KISS_HIP_E_HIP;
            for (int b = 0; b < 2 && rc == KISS_HIP_OK; b++)
                if (hipHostMalloc(&rd[t].pin[b], CHUNK, hipHostMallocDefault) != hipSuccess ||
                    hipEventCreateWithFlags(&rd[t].done[b], hipEventDisableTiming) != hipSuccess)
                    rc = KISS_HIP_E_NOMEM;
        }
        if (rc) break;
        const int device = ctx->device;
        auto work = [&](int t) {
            if (hipSetDevice(device) != hipSuccess) {
                status = KISS_HIP_E_HIP;
                return;
            }
            Reader &r = rd[t];
            bool used[2] = {false, false};
            int b = 0;
            for (uint64_t c = (uint64_t)t; c < chunks && status == KISS_HIP_OK; c += (uint64_t)readers) {
                const uint64_t off = c * CHUNK;
                const size_t want = bytes - off < CHUNK ? (size_t)(bytes - off) : CHUNK;
                if (used[b]) (void)hipEventSynchronize(r.done[b]); // the previous upload from this buffer has finished
                size_t got = 0;
                while (got < want) {
                    const ssize_t n = pread(fd, (char *)r.pin[b] + got, want - got, (off_t)(off + got));
                    if (n <= 0) {
                        status = KISS_HIP_E_IO;
                        return;
                    }
                    got += (size_t)n;
                }
                if (hipMemcpyAsync(d_raw + off, r.pin[b], want, hipMemcpyHostToDevice, r.stream) != hipSuccess) {
                    status = KISS_HIP_E_HIP;
                    return;
                }
                (void)hipEventRecord(r.done[b], r.stream);
                used[b] = true;
                b ^= 1;
            }
            if (hipStreamSynchronize(r.stream) != hipSuccess) status = KISS_HIP_E_HIP;
        };
        std::vector<std::thread> th; // (no exception crosses the C ABI: a reader that can not start is run by this thread)
        std::vector<int> orphan;
        for (int t = 1; t < readers; t++) {
            try {
                th.emplace_back(work, t);
            } catch (...) {
                orphan.push_back(t);
            }
        }
        work(0);
        for (int t : orphan) work(t);
        for (auto &x : th) x.join();
        rc = status;
        if (rc) break;
        rc = kiss_hip_ctx_parse_text_dev(ctx, d_raw, bytes, d_S, n_out, nullptr);
    } while (0);
    (void)hipStreamSynchronize(ctx->stream);
    close(fd);
    for (auto &r : rd) {
        for (int b = 0; b < 2; b++) {
            if (r.pin[b]) (void)hipHostFree(r.pin[b]);
            if (r.done[b]) (void)hipEventDestroy(r.done[b]);
        }
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    if (d_raw) (void)hipFree(d_raw);
    if (rc) {
        if (d_S) (void)hipFree(d_S);
        return rc;
    }
    *d_S_out = d_S;
    return KISS_HIP_OK;
}

int kiss_hip_alloc_dev(void **d_out, uint64_t bytes)
{
    if (!d_out) return KISS_HIP_E_INVALID;
    *d_out = nullptr;
    if (hipMalloc(d_out, bytes ? bytes : 1) != hipSuccess) return KISS_HIP_E_NOMEM;
    return KISS_HIP_OK;
}

int kiss_hip_free_dev(void *p)
{
    if (p && hipFree(p) != hipSuccess) return KISS_HIP_E_HIP;
    return KISS_HIP_OK;
}

int kiss_hip_copy_to_host(void *dst, const void *d_src, uint64_t bytes)
{
    if (bytes && (!dst || !d_src)) return KISS_HIP_E_INVALID;
    if (bytes && hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return KISS_HIP_E_HIP;
    return KISS_HIP_OK;
}

} // extern "C"
