// alias_table.hpp — O(1) discrete sampling table for light selection.
// Same construction as the reference's DiscreteSampler1D (src/alias_table.hpp:21-63): scale the weights to mean 1,
// split them into an "over-full" and an "under-full" stack in index order, then repeatedly top up the most recent
// under-full bucket from the most recent over-full one.  The resulting {prob, failId} pairs feed
// ImptSampData.q / .alias of the light records (scene.cpp:700-772).  Validated against the reference header
// compiled in place (oracle/kat/mint_kat.sh -> "alias_table" in tests/golden/kat_reference.json, tests/test_kat.py).
#pragma once
#include <vector>

namespace rth {

struct AliasBucket {
  float prob;   // probability of keeping the bucket's own index
  int failId;   // index taken otherwise
};

inline std::vector<AliasBucket> buildAliasTable(std::vector<float> w, float* sumOut = nullptr)
{
  const int n = int(w.size());
  float total = 0.f;
  for(float v : w) total += v;
  if(sumOut) *sumOut = total;
  const float scale = float(n) / total;
  for(float& v : w) v *= scale;

  std::vector<AliasBucket> table(n);
  std::vector<AliasBucket> over, under;  // LIFO
  over.reserve(n); under.reserve(n);
  for(int i = 0; i < n; i++) (w[i] > 1.f ? over : under).push_back(AliasBucket{w[i], i});

  while(!over.empty() && !under.empty()) {
    AliasBucket big = over.back(); over.pop_back();
    AliasBucket small = under.back(); under.pop_back();
    table[small.failId] = AliasBucket{small.prob, big.failId};
    big.prob -= (1.f - small.prob);
    (big.prob > 1.f ? over : under).push_back(big);
  }
  for(int i = int(over.size()) - 1; i >= 0; i--) table[over[i].failId] = over[i];
  for(int i = int(under.size()) - 1; i >= 0; i--) table[under[i].failId] = under[i];
  return table;
}

}  // namespace rth
