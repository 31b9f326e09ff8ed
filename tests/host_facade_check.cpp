// compiled (not run) by tests/test_abi.py: the C++ facade must satisfy the reference's SASorter shape
#include <concepts>
#include <string_view>
#include "../kiss_amd/csrc/host/kiss_hip_sorter.hpp"

// restatement of the reference concept (include/biovoltron/algo/sort/sorter.hpp:7-10) with a byte view
template <class T>
concept SASorterLike = requires(T t, std::basic_string_view<signed char> ref) { t.get_suffix_array_dna(ref); };
static_assert(SASorterLike<biovoltron::KissHipSorter<std::uint32_t>>);

int main(int argc, char**) {
  std::vector<std::uint8_t> S{0, 1, 2, 3, 0, 1};
  if (argc > 100) {  // never executed on a box without a GPU
    auto sa = biovoltron::KissHipSorter<>::get_suffix_array_dna(S, 256u, 1);
    auto sb = biovoltron::KissHipSorter2<>::get_suffix_array_dna(S);
    auto sc = biovoltron::KissHipSorter<>::get_suffix_array(std::string_view("general alphabet"), 256u, 1);
    return (int)(sa.size() + sb.size() + sc.size());
  }
  return 0;
}
