// nvdb_search -- prints the top-k of query 0 (reference apps/nvdb_search.cpp:6-41: same arguments,
// same lines).  Optional 4th argument "gpu" runs the scan on the MI355X.
#include <iomanip>
#include <iostream>
#include <string>

#include "nvdb/flat_index.h"
#include "nvdb/flat_index_hip.h"

int main(int argc, char** argv) {
  if (argc < 4) { std::cerr << "Usage: nvdb_search <base.vecbin> <query.vecbin> <k> [gpu]\n"; return 1; }
  const uint32_t k = static_cast<uint32_t>(std::stoul(argv[3]));
  nvdb::VectorDataset base, query;
  base.load(argv[1]);
  query.load(argv[2]);
  if (base.dim() != query.dim()) { std::cerr << "Dim mismatch: base.dim=" << base.dim() << ", query.dim=" << query.dim() << "\n"; return 2; }
  std::cout << "Base count=" << base.count() << " dim=" << base.dim() << " | Query count=" << query.count() << "\n";
  std::vector<nvdb::SearchResult> topk;
  if (argc >= 5 && std::string(argv[4]) == "gpu") topk = nvdb::FlatIndexHIP(&base).search_topk_dot(query.vector_ptr(0), k);
  else topk = nvdb::FlatIndex(&base).search_topk_dot(query.vector_ptr(0), k);
  std::cout << std::fixed << std::setprecision(6);
  for (size_t i = 0; i < topk.size(); ++i) std::cout << "#" << (i + 1) << " row=" << topk[i].id << " score=" << topk[i].score << "\n";
  return 0;
}
