// Host build of csrc/ttx_select.h for tests/test_select_host.py (g++, no GPU): the same source the kernels compile.
#include "ttx_select.h"
#include <vector>

extern "C" int ttx_host_topk1_index(const long long* values, int n) {
  std::vector<long long> v(values, values + n);
  std::vector<int> ix(n);
  return ttxsel::topk1_index(v.data(), ix.data(), n);
}
