// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or executed from the product path).
//
// Builds oracle/_ref/libkiss_ref.so from the REFERENCE'S OWN SOURCES where they lie under /root/reference
// (nothing of the reference is copied into this repository; the .so is git-ignored and travels to the GPU box as a
// binary, like our own built libraries).  What compiles from the reference unmodified, with plain g++ and no
// stand-in for anything the image lacks:
//
//   include/biovoltron/algo/sort/kiss_common.hpp   get_lms (:483-579), put_lms_suffix (:445-481),
//                                                  induced_sort = induced_L / induced_clear / induced_S (:14-443)
//   include/biovoltron/algo/sort/structs.hpp       ThreadState, PackedDNAString incl. load_prefix_length_125 (:122-169)
//                                                  and load_prefix_length_less_than_16 (:175-184)
//   include/biovoltron/algo/sort/utils.hpp, constant.hpp, container/xbit_vector.hpp
//
// What does NOT compile here: kiss1_core.hpp / kiss2_core.hpp / fm_index.hpp start with #include <spdlog/spdlog.h>
// (kiss1_core.hpp:6-7), an un-vendored submodule absent from /root/reference and from this image; writing a stand-in
// header is not allowed.  Hence the one stage of the pipeline that lives in kiss1_core.hpp,
// lms_suffix_direct_sort_dna (:24-145: 10-mer bucket scatter + per-bucket std::sort with the comparator lambda), is
// RESTATED below in kref_lms_sort() -- on top of the reference's own PackedDNAString loads and libstdc++'s std::sort,
// but with its control flow re-typed by us.  Everything else in kref_suffix_sort() is the reference's code executing.
//
// So this library pins, against the reference itself: the LMS list and the 5x256 histograms, the 10-mer prefix and the
// 125-base block loads the comparator is made of, LMS placement and both induction sweeps (with the reference's own
// OpenMP block scheduling).  It does not pin the comparator's control flow (depth rule, scalar tail, tie rules):
// oracle/kiss_oracle.c and this file restate kiss1_core.hpp:94-135 independently of each other (C with byte compares
// there, AVX2 block compares here) and tests/test_ref_pin.py checks that the two agree.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <immintrin.h>
#include <omp.h>

#include <biovoltron/algo/sort/kiss_common.hpp>
#include <biovoltron/algo/sort/structs.hpp>

namespace bk = biovoltron::kiss;
using u8 = uint8_t;
using u32 = uint32_t;
using States = bk::vector<bk::ThreadState<u32>>;
using Packed = bk::PackedDNAString<u8, u32>;

namespace {

// Order of two 125-base blocks as the reference's packed representation defines it: byte 31 of the 256-bit word is the
// most significant one (reverse packing, structs.hpp:88-116).  Returns <0, 0, >0.
inline int block125_order(__m256i a, __m256i b)
{
    const unsigned ne = ~(unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, b));
    if (!ne) return 0;
    const unsigned top = 31u - (unsigned)__builtin_clz(ne);
    const unsigned a_ge_b = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_max_epu8(a, b), a));
    return ((a_ge_b >> top) & 1u) ? 1 : -1;
}

// Restatement of the comparator lambda of lms_suffix_direct_sort_dna (kiss1_core.hpp:94-135); see the file header.
struct KOrderLess {
    const u8 *S;
    u32 n, k;
    Packed *P;
    bool operator()(u32 i, u32 j) const
    {
        constexpr u32 B = bk::KISS1_SPLIT_SORT_STRIDE_DNA;
        uint64_t done = 0; // the reference counts in size_type; k = 2^32-1 never lets `done <= k` fail before the text ends
        uint64_t a = i, b = j;
        while (done <= k && a + B <= n && b + B <= n) {
            const int c = block125_order(P->load_prefix_length_125((u32)a), P->load_prefix_length_125((u32)b));
            if (c) return c < 0;
            done += B, a += B, b += B;
        }
        for (; done + 1 <= k && a < n && b < n; done++, a++, b++)
            if (S[a] != S[b]) return S[a] < S[b];
        if (done >= k) return a < b;
        return a == n;
    }
};

// Restatement of lms_suffix_direct_sort_dna's phases (kiss1_core.hpp:41-144): stable counting sort of the ascending
// LMS list on the reference's own 10-mer prefix load, then std::sort of every bucket with the comparator.
void kref_lms_sort(const bk::vector<u8> &S, u32 *lms, u32 m, u32 k, int T, States &states)
{
    Packed P(S, states, (size_t)T);
    constexpr size_t NB = bk::KISS1_SPLIT_SORT_BUCKET_SIZE;
    std::vector<u32> start(NB + 1, 0), out(m);
    std::vector<u32> pre(m);
#pragma omp parallel for num_threads(T)
    for (long long i = 0; i < (long long)m; i++)
        pre[i] = P.load_prefix_length_less_than_16(lms[i], bk::KISS1_SPLIT_SORT_PREFIX_SIZE);
    for (u32 i = 0; i < m; i++) start[pre[i] + 1]++;
    for (size_t b = 0; b < NB; b++) start[b + 1] += start[b];
    {
        std::vector<u32> cur(start.begin(), start.end() - 1);
        for (u32 i = 0; i < m; i++) out[cur[pre[i]]++] = lms[i];
    }
    KOrderLess less{S.data(), (u32)S.size(), k, &P};
#pragma omp parallel for schedule(dynamic) num_threads(T)
    for (long long b = 0; b < (long long)NB; b++)
        std::sort(out.begin() + start[b], out.begin() + start[b + 1], less);
    std::memcpy(lms, out.data(), sizeof(u32) * (size_t)m);
}

} // namespace

extern "C" {

int kref_max_threads(void) { return omp_get_max_threads(); }

// get_lms (kiss_common.hpp:543-579): ascending LMS positions + the sentinel n -> lms_out[0..m), returns m;
// hist (optional) = the 5 x 256 per-thread buckets summed over the threads.
u32 kref_get_lms(const u8 *S_, u32 n, int T, u32 *lms_out, u32 *hist)
{
    bk::vector<u8> S(S_, S_ + n);
    bk::vector<u32> SA((size_t)n + 1);
    States states((size_t)T);
    const u32 m = bk::get_lms(S, SA, states);
    std::memcpy(lms_out, SA.data(), sizeof(u32) * (size_t)m);
    if (hist) {
        std::memset(hist, 0, sizeof(u32) * 5 * bk::CHAR_SIZE);
        for (auto &st : states)
            for (size_t i = 0; i < 5 * bk::CHAR_SIZE; i++) hist[i] += st.bucket[i];
    }
    return m;
}

// PackedDNAString::load_prefix_length_less_than_16(idx, 10) (structs.hpp:175-184) for cnt positions
void kref_prefix10(const u8 *S_, u32 n, int T, const u32 *idx, u32 cnt, u32 *out)
{
    bk::vector<u8> S(S_, S_ + n);
    States states((size_t)T);
    Packed P(S, states, (size_t)T);
    for (u32 i = 0; i < cnt; i++) out[i] = P.load_prefix_length_less_than_16(idx[i], bk::KISS1_SPLIT_SORT_PREFIX_SIZE);
}

// PackedDNAString::load_prefix_length_125(idx) (structs.hpp:122-169): the raw 32 bytes, for cnt positions with
// idx + 125 <= n (the comparator's own precondition, kiss1_core.hpp:96-98)
void kref_load125(const u8 *S_, u32 n, int T, const u32 *idx, u32 cnt, u8 *out32)
{
    bk::vector<u8> S(S_, S_ + n);
    States states((size_t)T);
    Packed P(S, states, (size_t)T);
    for (u32 i = 0; i < cnt; i++)
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(out32 + 32 * (size_t)i), P.load_prefix_length_125(idx[i]));
}

// The call sequence of kiss1_suffix_array_dna (kiss1_core.hpp:229-268) with the reference's get_lms, put_lms_suffix
// and induced_sort; the LMS order either comes from the caller (sorted_lms_in: m entries, sentinel first) or from
// kref_lms_sort above.  SA_out: n + 1 entries.  lms_sorted_out (optional): the m LMS positions in k-order.
int kref_suffix_sort(const u8 *S_, u32 n, u32 k, int T, const u32 *sorted_lms_in, u32 *SA_out, u32 *lms_sorted_out,
                     u32 *m_out)
{
    if (n == 0) { // kiss1_core.hpp:237-238
        SA_out[0] = 0;
        if (m_out) *m_out = 0;
        return 0;
    }
    bk::vector<u8> S(S_, S_ + n);
    bk::vector<u32> SA((size_t)n + 1);
    States states((size_t)T);
    const u32 m = bk::get_lms(S, SA, states);
    if (sorted_lms_in) std::memcpy(SA.data(), sorted_lms_in, sizeof(u32) * (size_t)m);
    else kref_lms_sort(S, SA.data(), m, k, T, states);
    if (SA[0] != n) return -1; // the sentinel sorts first (kiss1_core.hpp:263: SA1 = SA[1..m))
    if (lms_sorted_out) std::memcpy(lms_sorted_out, SA.data(), sizeof(u32) * (size_t)m);
    if (m_out) *m_out = m;
    auto SA1 = std::ranges::subrange(SA.begin() + 1, SA.begin() + m);
    bk::put_lms_suffix(S, SA, SA1, states);
    bk::induced_sort(S, SA, states);
    std::memcpy(SA_out, SA.data(), sizeof(u32) * ((size_t)n + 1));
    return 0;
}

} // extern "C"
