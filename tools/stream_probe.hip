// tools/stream_probe.hip — what HBM delivers for the ACCESS PATTERN of the shade kernel, with no arithmetic:
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe
// N slots; R separately allocated arrays of 16-byte records read per slot and W of them written back (the RGB k_logic reads
// 8 x 16 B + 2 x 4 B and writes 8 x 16 B + 1 x 4 B per live slot), 256-thread workgroups, one lane per slot, all loads
// issued before the first use.  Prints GB/s for a few (R, W) pairs and for a plain 1-array copy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Arrays { float4* a[12]; };

template <int R, int W>
__global__ __launch_bounds__(256) void probe(Arrays arr, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = arr.a[k][i];
    float4 s = v[0];
#pragma unroll
    for (int k = 1; k < R; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
#pragma unroll
    for (int k = 0; k < W; ++k) arr.a[k][i] = make_float4(s.x + k, s.y, s.z, s.w);
}

template <int R, int W>
static int run(Arrays arr, uint32_t n) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((probe<R, W>), dim3((n + 255) / 256), dim3(256), 0, 0, arr, n);
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((probe<R, W>), dim3((n + 255) / 256), dim3(256), 0, 0, arr, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n * 16.0 * (R + W);
    printf("read %2d + write %2d arrays of %u x 16 B: %7.1f us per launch, %7.1f GB/s\n", R, W, n, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    return 0;
}

int main() {
    const uint32_t n = 7372800;          // the slot count of the headline run
    Arrays arr;
    for (int k = 0; k < 12; ++k) {
        void* p = nullptr;
        CHECK(hipMalloc(&p, (size_t)n * 16 + 65536));
        CHECK(hipMemset(p, 0, (size_t)n * 16));
        arr.a[k] = (float4*)((char*)p + (k % 7) * 256 + (k * 4352) % 61440);     // skewed bases like DevArray
    }
    if (run<1, 1>(arr, n)) return 1;
    if (run<4, 4>(arr, n)) return 1;
    if (run<9, 8>(arr, n)) return 1;      // the shade kernel's stream count
    if (run<9, 0>(arr, n)) return 1;
    if (run<1, 8>(arr, n)) return 1;
    if (run<5, 10>(arr, n)) return 1;     // the regen kernel's, if it were dense
    return 0;
}
