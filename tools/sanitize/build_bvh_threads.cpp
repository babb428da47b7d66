#include "../../include/rvpt_hip.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
int main() {
    const size_t n = 120000;  // above the multi-thread threshold
    std::vector<rvpt_triangle> t(n);
    srand(1);
    for (auto &x : t) { float c[3] = {rand() / 1e7f, rand() / 1e7f, rand() / 1e7f}; for (int k = 0; k < 3; ++k) { x.vert0[k] = c[k]; x.vert1[k] = c[k] + (k == 0); x.vert2[k] = c[k] + (k == 1); } }
    std::vector<rvpt_bvh_node> nodes(2 * n - 1); std::vector<uint32_t> idx(n); size_t nn = 0;
    int rc = rvpt_bvh_build(t.data(), n, nodes.data(), &nn, idx.data());
    printf("rc %d nodes %zu\n", rc, nn);
    return rc;
}
