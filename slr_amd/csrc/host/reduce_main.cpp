// reduce_main.cpp — a C++ host exercising the multi-GPU entry points of the C ABI the way a libSLR host with N devices would:
// one rank per device, each rendering its tile shard (slrhip_render_begin's shard) and all of them calling
// slrhip_reduce_framebuffer over an RCCL communicator.  With the devices this process sees (a 1-GPU box: one rank, a
// communicator of size 1) the ranks run one after the other inside one process through ncclCommInitAll + ncclGroupStart/End;
// the sum of the shards must equal the unsharded frame bit for bit.
//   usage: reduce_main [numShards] [width] [height] [spp] [tables.bin]
//          reduce_main --ranks N [width] [height] [spp] [tables.bin]
// --ranks N is the flow of an N-GPU node in C++: the parent forks N children BEFORE any process has touched the GPU, rank r takes
// device r, rank 0 creates the ncclUniqueId and hands it to the others over pipes, every rank renders the tiles t % N == r and
// all of them meet in ONE slrhip_reduce_framebuffer (ncclReduce over xGMI) on rank 0, which checks the assembled frame against
// its own unsharded render bit for bit.  (On a one-GPU box only N = 1 can run; RCCL refuses two ranks on one device.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "SLRHip.h"
#include "cornell_scene.h"

#define CHECK(expr) do { int rc_ = (expr); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, slrhip_last_error_string()); return 1; } } while (0)
#define HIPCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 1; } } while (0)


// One rank of the N-process flow (runs in a child; the first GPU call of the process is in here).
static int runRank(int rank, int ranks, int idReadFd, const std::vector<int>& idWriteFds, int W, int H, uint32_t spp, const char* tables) {
    ncclUniqueId id;
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) { std::fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
        for (int fd : idWriteFds)
            if (write(fd, &id, sizeof(id)) != (ssize_t)sizeof(id)) { std::fprintf(stderr, "rank 0: could not hand the id over\n"); return 1; }
    }
    else if (read(idReadFd, &id, sizeof(id)) != (ssize_t)sizeof(id)) { std::fprintf(stderr, "rank %d: no ncclUniqueId from rank 0\n", rank); return 1; }
    HIPCHECK(hipSetDevice(rank));
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, ranks, id, rank) != ncclSuccess) { std::fprintf(stderr, "rank %d: ncclCommInitRank failed\n", rank); return 1; }
    SLRHip::Scene scene;
    if (!scene.loadSpectralTables(tables)) return 1;
    cornell::build(scene, W, H, false);
    slrhip_scene_desc desc = scene.desc();
    slrhip_render_settings st = {W, H, 0.0f, 0.0f, 1.0f, 1509761209};
    slrhip_config cfg = {rank, SLRHIP_MODE_RGB, 4, 0};      // a fixed stripe count: the shards then sum to the full frame bit for bit
    const size_t n = (size_t)W * H * 3;
    hipStream_t stream;
    HIPCHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    float* dFull = nullptr;
    if (rank == 0) HIPCHECK(hipMalloc(&dFull, n * sizeof(float)));
    slrhip_ctx* ctx = nullptr;
    CHECK(slrhip_create(&cfg, &ctx));
    CHECK(slrhip_upload_scene(ctx, &desc));
    slrhip_shard sh = {(uint32_t)rank, (uint32_t)ranks};
    CHECK(slrhip_render_begin(ctx, &st, sh));
    CHECK(slrhip_render(ctx, 0, spp, stream));
    CHECK(slrhip_reduce_framebuffer(ctx, comm, 0, dFull, n, stream));      // the one exchange of the path
    HIPCHECK(hipStreamSynchronize(stream));
    int rc = 0;
    if (rank == 0) {
        std::vector<float> got(n), full(n);
        HIPCHECK(hipMemcpy(got.data(), dFull, n * sizeof(float), hipMemcpyDeviceToHost));
        slrhip_shard whole = {0, 1};
        CHECK(slrhip_render_begin(ctx, &st, whole));
        CHECK(slrhip_render(ctx, 0, spp, stream));
        CHECK(slrhip_read_framebuffer(ctx, full.data(), n));
        size_t bad = 0;
        double sum = 0.0;
        for (size_t i = 0; i < n; ++i) { if (std::memcmp(&got[i], &full[i], 4) != 0 && !(got[i] == 0.0f && full[i] == 0.0f)) ++bad; sum += full[i]; }
        int count = 0;
        ncclCommCount(comm, &count);
        std::printf("reduce_main: %d ranks (communicator size %d), %zu floats, %zu differ, frame sum %.6g\n", ranks, count, n, bad, sum);
        rc = bad == 0 && sum > 0.0 && count == ranks ? 0 : 2;
        std::fflush(stdout);                                   // the child leaves through _exit
    }
    slrhip_destroy(ctx);
    ncclCommDestroy(comm);
    return rc;
}

static int runRanks(int ranks, int W, int H, uint32_t spp, const char* tables) {
    if (ranks < 1 || ranks > 64) { std::fprintf(stderr, "--ranks: 1 .. 64\n"); return 1; }
    // pipes rank 0 -> rank r for the ncclUniqueId; children are forked before this process makes any GPU call
    std::vector<int> readFd(ranks, -1), writeFd;
    for (int r = 1; r < ranks; ++r) {
        int fds[2];
        if (pipe(fds) != 0) { std::perror("pipe"); return 1; }
        readFd[r] = fds[0];
        writeFd.push_back(fds[1]);
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) _exit(runRank(r, ranks, readFd[r], r == 0 ? writeFd : std::vector<int>(), W, H, spp, tables));
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t k : kids) {
        int status = 0;
        waitpid(k, &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = WIFEXITED(status) ? WEXITSTATUS(status) : 3;
    }
    return rc;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::strcmp(argv[1], "--ranks") == 0)
        return runRanks(std::atoi(argv[2]), argc > 3 ? std::atoi(argv[3]) : 160, argc > 4 ? std::atoi(argv[4]) : 120, argc > 5 ? (uint32_t)std::atoi(argv[5]) : 8,
                        argc > 6 ? argv[6] : "slr_amd/data/upsampling_tables.bin");
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
