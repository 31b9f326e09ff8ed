// The header-only C++ facade a maintainer drops into the reference.
//   no argument : compile/link check only (tests/test_abi.py, runs on a box without a GPU): exits 0 without device work
//   "run" FILE  : really calls KissHipSorter / KissHipSorter2 / get_suffix_array on a GPU box and writes the four
//                 suffix arrays to FILE (u32 LE, concatenated) for tests/test_cli_gpu.py to compare with the ctypes path
#include <concepts>
#include <cstdio>
#include <cstring>
#include <string_view>
#include "../kiss_amd/csrc/host/kiss_hip_sorter.hpp"

// restatement of the reference concept (include/biovoltron/algo/sort/sorter.hpp:7-10) with a byte view
template <class T>
concept SASorterLike = requires(T t, std::basic_string_view<signed char> ref) { t.get_suffix_array_dna(ref); };
static_assert(SASorterLike<biovoltron::KissHipSorter<std::uint32_t>>);
static_assert(SASorterLike<biovoltron::KissHipSorter<std::uint64_t>>);  // the reference's template takes a size_type (kiss1_sorter.hpp:7)

int main(int argc, char** argv) {
  if (argc < 3 || std::strcmp(argv[1], "run") != 0) return 0;
  // deterministic text: 50 000 bases from a 64-bit LCG, with a planted 600-base copy (ties beyond k = 256)
  std::vector<std::uint8_t> S(50000);
  std::uint64_t x = 12345;
  for (auto& c : S) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    c = (std::uint8_t)(x >> 62);
  }
  for (int i = 0; i < 600; i++) S[30000 + i] = S[1000 + i];
  try {
    auto sa = biovoltron::KissHipSorter<>::get_suffix_array_dna(S, 256u, 1);
    auto sb = biovoltron::KissHipSorter2<>::get_suffix_array_dna(S);
    std::string text(S.size(), 'A');
    for (std::size_t i = 0; i < S.size(); i++) text[i] = (char)('A' + S[i]);
    auto sc = biovoltron::KissHipSorter<>::get_suffix_array(std::string_view(text), 256u, 1);
    // the same sort sharded over two shares of device 0 by this process (what `kiss suffix_sort --gpus N` selects)
    biovoltron::KissHipSorter<>::set_devices({0, 0});
    auto sd = biovoltron::KissHipSorter<>::get_suffix_array_dna(S, 256u, 1);
    biovoltron::KissHipSorter<>::set_devices({});
    // 64-bit size_type: the same values, widened
    auto s64 = biovoltron::KissHipSorter<std::uint64_t>::get_suffix_array_dna(S, 256u, 1);
    if (s64.size() != sa.size()) return 5;
    for (std::size_t i = 0; i < sa.size(); i++)
      if (s64[i] != sa[i]) return 5;
    if (sa.size() != S.size() + 1 || sb.size() != S.size() + 1 || sc.size() != S.size() + 1 || sd.size() != S.size() + 1) return 3;
    std::FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 4;
    std::fwrite(sa.data(), 4, sa.size(), f);
    std::fwrite(sb.data(), 4, sb.size(), f);
    std::fwrite(sc.data(), 4, sc.size(), f);
    std::fwrite(sd.data(), 4, sd.size(), f);
    std::fclose(f);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "host_facade_check: %s\n", e.what());
    return 2;
  }
  return 0;
}
