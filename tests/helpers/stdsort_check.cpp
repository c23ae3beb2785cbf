// Test helper: checks csrc/stdsort_replica.h against the real libstdc++ std::sort on the
// same container type and comparator the reference uses (rtpose.cpp:144-152, 953-954).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../caffe_rtpose_amd/csrc/stdsort_replica.h"

struct ColumnCompare {
  bool operator()(const std::vector<double>& l, const std::vector<double>& r) const { return l[2] > r[2]; }
};
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
  int trials = argc > 1 ? atoi(argv[1]) : 2000;
  long checked = 0;
  for (int t = 0; t < trials; ++t) {
    int n = (int)(next() % (t % 10 == 0 ? 4097 : 300));
    int levels = 1 + (int)(next() % (t % 3 == 0 ? 3 : 1000));   // few distinct keys => many ties
    int pattern = (int)(next() % 5);
    std::vector<std::vector<double>> ref(n, std::vector<double>(4, 0));
    std::vector<rtp::Cand> mine(n);
    for (int i = 0; i < n; ++i) {
      float sc;
      if (pattern == 0) sc = (float)(next() % levels) / levels;
      else if (pattern == 1) sc = (float)i / (n + 1);            // ascending
      else if (pattern == 2) sc = (float)(n - i) / (n + 1);      // descending
      else if (pattern == 3) sc = 1.0f;                           // all equal
      else sc = (float)((i * 7919) % levels) / levels;            // organ-pipe-ish
      ref[i][0] = i / 64 + 1; ref[i][1] = i % 64 + 1; ref[i][2] = sc;
      mine[i].score = sc; mine[i].ij = ((i / 64 + 1) << 16) | (i % 64 + 1);
    }
    if (!ref.empty()) std::sort(ref.begin(), ref.end(), ColumnCompare());
    rtp::std_sort_replica(mine.data(), n);
    for (int i = 0; i < n; ++i) {
      int ij = ((int)ref[i][0] << 16) | (int)ref[i][1];
      if (ij != mine[i].ij || (float)ref[i][2] != mine[i].score) {
        printf("MISMATCH trial %d n %d pattern %d at %d\n", t, n, pattern, i);
        return 1;
      }
    }
    checked += n;
  }
  printf("OK %d trials %ld elements\n", trials, checked);
  return 0;
}
