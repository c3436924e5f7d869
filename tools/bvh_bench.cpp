// tools/bvh_bench.cpp — times slr_amd/csrc/bvh.cpp on a synthetic n x n displaced grid (host only, no GPU).
//   g++ -O3 -std=c++17 -I include tools/bvh_bench.cpp slr_amd/csrc/bvh.cpp -o /tmp/bvh_bench -lpthread && /tmp/bvh_bench 2236
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../slr_amd/csrc/bvh.h"

int main(int argc, char** argv) {
    uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
    std::vector<slrhip_vertex> v((size_t)(n + 1) * (n + 1));
    std::vector<slrhip_triangle> t((size_t)2 * n * n);
    for (uint32_t z = 0; z <= n; ++z)
        for (uint32_t x = 0; x <= n; ++x) {
            slrhip_vertex& p = v[(size_t)z * (n + 1) + x];
            p.position[0] = -4.0f + 8.0f * x / n;
            p.position[2] = -4.0f + 8.0f * z / n;
            p.position[1] = 0.3f * std::sin(0.01f * x) * std::cos(0.013f * z) + 0.002f * ((x * 2654435761u ^ z * 40503u) >> 24) / 256.0f;
        }
    for (uint32_t z = 0; z < n; ++z)
        for (uint32_t x = 0; x < n; ++x) {
            uint32_t a = z * (n + 1) + x, b = a + 1, c = a + n + 2, d = a + n + 1;
            slrhip_triangle* q = &t[2 * ((size_t)z * n + x)];
            q[0].v[0] = a; q[0].v[1] = d; q[0].v[2] = c; q[0].material = 0;
            q[1].v[0] = a; q[1].v[1] = c; q[1].v[2] = b; q[1].material = 0;
        }
    auto t0 = std::chrono::steady_clock::now();
    slrhip::QBVH bvh;
    int rc = slrhip::buildQBVH(v.data(), t.data(), (uint32_t)t.size(), &bvh);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("rc %d tris %zu nodes %zu leafTris %zu depth %u  %.2f s\n", rc, t.size(), bvh.nodes.size(), bvh.leafTris.size(), bvh.depth, s);
    return rc;
}
