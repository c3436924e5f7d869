// reduce_main.cpp — a C++ host exercising the multi-GPU entry points of the C ABI the way a libSLR host with N devices would:
// one rank per device, each rendering its tile shard (slrhip_render_begin's shard) and all of them calling
// slrhip_reduce_framebuffer over an RCCL communicator.  With the devices this process sees (a 1-GPU box: one rank, a
// communicator of size 1) the ranks run one after the other inside one process through ncclCommInitAll + ncclGroupStart/End;
// the sum of the shards must equal the unsharded frame bit for bit.
//   usage: reduce_main [numShards] [width] [height] [spp] [tables.bin]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "SLRHip.h"
#include "cornell_scene.h"

#define CHECK(expr) do { int rc_ = (expr); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, slrhip_last_error_string()); return 1; } } while (0)
#define HIPCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 1; } } while (0)


int main(int argc, char** argv) {
    const uint32_t shards = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 4;
    const int W = argc > 2 ? std::atoi(argv[2]) : 160, H = argc > 3 ? std::atoi(argv[3]) : 120;
    const uint32_t spp = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 8;
    SLRHip::Scene scene;
    if (!scene.loadSpectralTables(argc > 5 ? argv[5] : "slr_amd/data/upsampling_tables.bin")) return 1;
    cornell::build(scene, W, H, false);
    slrhip_scene_desc desc = scene.desc();
    slrhip_render_settings st = {W, H, 0.0f, 0.0f, 1.0f, 1509761209};
    slrhip_config cfg = {0, SLRHIP_MODE_RGB, 4, 0};      // a fixed stripe count: the shards then sum to the full frame bit for bit
    const size_t n = (size_t)W * H * 3;

    int dev = 0;
    ncclComm_t comm;
    if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    hipStream_t stream;
    HIPCHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    float* dAcc = nullptr;
    float* dPart = nullptr;
    HIPCHECK(hipMalloc(&dAcc, n * sizeof(float)));
    HIPCHECK(hipMalloc(&dPart, n * sizeof(float)));

    slrhip_ctx* ctx = nullptr;
    CHECK(slrhip_create(&cfg, &ctx));
    CHECK(slrhip_upload_scene(ctx, &desc));
    // the unsharded frame
    slrhip_shard whole = {0, 1};
    CHECK(slrhip_render_begin(ctx, &st, whole));
    CHECK(slrhip_render(ctx, 0, spp, stream));
    std::vector<float> full(n), total(n, 0.0f), part(n);
    CHECK(slrhip_read_framebuffer(ctx, full.data(), n));
    // every shard through render -> slrhip_reduce_framebuffer (communicator of this process's one device), summed on the host
    for (uint32_t r = 0; r < shards; ++r) {
        slrhip_shard sh = {r, shards};
        CHECK(slrhip_render_begin(ctx, &st, sh));
        CHECK(slrhip_render(ctx, 0, spp, stream));
        CHECK(slrhip_reduce_framebuffer(ctx, comm, 0, dPart, n, stream));
        HIPCHECK(hipStreamSynchronize(stream));
        HIPCHECK(hipMemcpy(part.data(), dPart, n * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            if (part[i] != 0.0f && total[i] != 0.0f) { std::fprintf(stderr, "shards overlap at float %zu\n", i); return 1; }
            total[i] += part[i];
        }
    }
    size_t bad = 0;
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) { if (std::memcmp(&total[i], &full[i], 4) != 0 && !(total[i] == 0.0f && full[i] == 0.0f)) ++bad; sum += full[i]; }
    slrhip_destroy(ctx);
    ncclCommDestroy(comm);
    std::printf("reduce_main: %u shards, %zu floats, %zu differ, frame sum %.6g\n", shards, n, bad, sum);
    return bad == 0 && sum > 0.0 ? 0 : 2;
}
