// kiss_hip_sorter.hpp -- header-only C++ host facade over the C ABI (include/kiss_hip.h).
//
// `biovoltron::KissHipSorter<uint32_t>` has the shape of the reference's sorter facades
// (reference include/biovoltron/algo/sort/kiss1_sorter.hpp:8-50, kiss2_sorter.hpp:8-50) and satisfies the
// `SASorter` concept (algo/sort/sorter.hpp:7-10): static `get_suffix_array_dna(S, k, num_threads)` returning an
// SA of n+1 entries, the range overload, and `prepare_aligned_ref`.  It can therefore be the `Sorter` template
// argument of `FMIndex<SA_INTV, size_type, Sorter>` (algo/align/exact_match/fm_index.hpp:384-387) and an
// alternative of the std::variant in `suffix_sort_main` (include/command/suffix_sort.hpp:37-60).
// Errors become exceptions here, like the reference's own error style (std::invalid_argument / bad_alloc);
// nothing C++ crosses the ABI itself.  `num_threads` is accepted and ignored (no result depends on it).
#pragma once
#include <cstdint>
#include <mutex>
#include <new>
#include <ranges>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/kiss_hip.h"

namespace biovoltron {

template <typename size_type = std::uint32_t>
struct KissHipSorter {
  // The device computes 32-bit indexes (the CLI's index type, suffix_sort.hpp:37; n <= KISS_HIP_MAX_N < 2^32).  A wider
  // size_type (the reference's template takes any unsigned type, kiss1_sorter.hpp:7) gets the same values widened.
  static_assert(sizeof(size_type) == 4 || sizeof(size_type) == 8, "size_type: uint32_t or uint64_t");
  using SA_t = std::vector<size_type>;

  // runs `call(uint32_t* SA)` and returns the suffix array as SA_t
  template <class F>
  static SA_t with_u32(std::size_t n, F&& call) {
    if constexpr (sizeof(size_type) == 4) {
      SA_t SA(n + 1);
      call(reinterpret_cast<std::uint32_t*>(SA.data()));
      return SA;
    } else {
      std::vector<std::uint32_t> narrow(n + 1);
      call(narrow.data());
      return SA_t(narrow.begin(), narrow.end());
    }
  }

  // Which GPU(s) the static facade sorts on: process-wide configuration like the reference's own globals (tbb::global_control,
  // the default logger), but safe to set from one thread while others sort -- a call takes a copy under the lock.
  //   set_device(d)         one GPU (default: device 0)
  //   set_devices({0,1,..}) the LMS sort sharded over these GPUs of the node by this process
  //                         (kiss_hip_suffix_sort_dna_u32_multi; the induction runs on the first one) -- what
  //                         `kiss suffix_sort --gpus N` selects; an empty list goes back to one GPU
  struct Config {
    int device = 0;
    std::vector<int> devices;
  };
  static void set_device(int d) {
    std::lock_guard<std::mutex> lock(config_mutex());
    config_ref().device = d;
  }
  static void set_devices(std::vector<int> d) {
    std::lock_guard<std::mutex> lock(config_mutex());
    config_ref().devices = std::move(d);
  }
  static Config config() {
    std::lock_guard<std::mutex> lock(config_mutex());
    return config_ref();
  }
  static int device() { return config().device; }
  static std::vector<int> devices() { return config().devices; }

 private:
  static std::mutex& config_mutex() {
    static std::mutex m;
    return m;
  }
  static Config& config_ref() {
    static Config c;
    return c;
  }

 public:
  static void check(int rc, const char* where) {
    if (rc == KISS_HIP_OK) return;
    if (rc == KISS_HIP_E_NOMEM) throw std::bad_alloc{};
    throw std::runtime_error(std::string(where) + ": " + kiss_hip_strerror(rc));
  }

  // kiss1_sorter.hpp:46-49 (the reference copies into a 64-byte aligned vector; the ABI has no alignment need)
  static auto prepare_aligned_ref(const std::ranges::random_access_range auto& ref) {
    std::vector<std::uint8_t> S(std::ranges::size(ref));
    std::size_t i = 0;
    for (auto c : ref) S[i++] = static_cast<std::uint8_t>(c);
    return S;
  }

  // kiss1_sorter.hpp:20-26
  static SA_t get_suffix_array_dna(const std::vector<std::uint8_t>& S, size_type k = 256u,
                                   std::size_t /*num_threads*/ = std::thread::hardware_concurrency(),
                                   int algo = KISS_HIP_ALGO_PARALLEL_SORTING) {
    // an order beyond 32 bits is the unbounded order (the CLI truncates -1 to 0xFFFFFFFF, suffix_sort.hpp:35-37)
    const std::uint32_t k32 = static_cast<std::uint64_t>(k) > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<std::uint32_t>(k);
    const Config cfg = config();
    return with_u32(S.size(), [&](std::uint32_t* SA) {
      if (!cfg.devices.empty())
        check(kiss_hip_suffix_sort_dna_u32_multi(S.data(), S.size(), k32, algo, SA, cfg.devices.data(),
                                                 static_cast<int>(cfg.devices.size())),
              "kiss_hip_suffix_sort_dna_u32_multi");
      else
        check(kiss_hip_suffix_sort_dna_u32(S.data(), S.size(), k32, algo, SA, cfg.device), "kiss_hip_suffix_sort_dna_u32");
    });
  }

  // range overload, kiss1_sorter.hpp:10-18
  static SA_t get_suffix_array_dna(const std::ranges::random_access_range auto& ref, size_type k = 256u,
                                   std::size_t num_threads = std::thread::hardware_concurrency()) {
    return get_suffix_array_dna(prepare_aligned_ref(ref), k, num_threads);
  }

  // general alphabet (bytes), kiss1_sorter.hpp:28-45 -> kiss1_suffix_array (kiss1_core.hpp:270-311): the reference
  // defines only the k-order property there; this returns the exact suffix array, which has it for every k
  static SA_t get_suffix_array(const std::ranges::random_access_range auto& ref, size_type /*k*/ = 256u,
                               std::size_t /*num_threads*/ = std::thread::hardware_concurrency()) {
    const auto S = prepare_aligned_ref(ref);
    return with_u32(S.size(), [&](std::uint32_t* SA) {
      check(kiss_hip_suffix_sort_u8(S.data(), S.size(), SA, device()), "kiss_hip_suffix_sort_u8");
    });
  }
};

// KISS2 (PREFIX_DOUBLING): k >= n gives the exact suffix array by rank doubling; a bounded k gives the same
// k-ordered array as KISS1 (the reference's bounded-k KISS2 output is thread-count dependent), see DESIGN.md section 8
template <typename size_type = std::uint32_t>
struct KissHipSorter2 : KissHipSorter<size_type> {
  using SA_t = typename KissHipSorter<size_type>::SA_t;
  static SA_t get_suffix_array_dna(const std::vector<std::uint8_t>& S, size_type k = static_cast<size_type>(0xFFFFFFFFu),
                                   std::size_t t = std::thread::hardware_concurrency()) {
    return KissHipSorter<size_type>::get_suffix_array_dna(S, k, t, KISS_HIP_ALGO_PREFIX_DOUBLING);
  }
};

}  // namespace biovoltron
