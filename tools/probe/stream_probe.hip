// Probe (gfx950): does the LAYOUT of the latent attention's key stream cost HBM bandwidth?  The cross launch reads, per
// decode row, one crop's 197 x 1,536 B = 302,592 contiguous bytes, 48 KiB (32 keys) per ring slot: 256 blocks sweep 256
// far-apart streams.  A copy kernel, whose waves sweep one region together, reaches 6.3 TB/s; the bare ring of the
// attention's pattern saturated at 5.4.  Here: the same bare ring (3 slots of 48 KiB, three DMA waves,
// nothing consumed), rows r = block, block + 256, ..., in two layouts of the same bytes:
//   A  row-major        : tile t of row r at  r * ROW + t * TILE                    (what the engine has)
//   B  tile-interleaved : tile t of row r at ((r / G) * T + t) * G * TILE + (r % G) * TILE   (G rows' tile t adjacent)
//     hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/stream_probe tools/probe/stream_probe.hip && tools/probe/bin/stream_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16_nt(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 2);
}

constexpr int TILE = 48 * 1024;

// layout: 0 = A, 1 = B.  tiles per row T (7 for 197 keys, the last one short in the engine; whole here).
__global__ __launch_bounds__(256, 1) void stream_k(const char* x, int rows, int T, int G, int layout, unsigned long long* rt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    int g = 0;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        for (int t = 0; t < T; ++t, ++g) {
            const size_t off = layout == 0 ? ((size_t)r * T + t) * TILE : (((size_t)(r / G) * T + t) * G + (r % G)) * (size_t)TILE;
            if (wave < 3) {
                char* slot = smem + (g % 3) * TILE;
#pragma unroll
                for (int i = 0; i < 16; ++i) glds16_nt(x + off + (size_t)(wave + 3 * i) * 1024 + lane * 16, slot + (wave + 3 * i) * 1024);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // two tiles stay in flight
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) rt[blockIdx.x] = r1 - r0;
}

static void run(const char* x, int blocks, int rows, int T, int G, int layout, unsigned long long* dr, const char* what) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_k), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * TILE));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_k, dim3(blocks), dim3(256), 3 * TILE, 0, x, rows, T, G, layout, dr);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const double bytes = (double)rows * T * TILE;
    printf("%-44s rows %5d: %8.1f us -> %5.2f TB/s (%.3f of 8 TB/s)\n", what, rows, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 8e12);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount, T = 7, rows = 2560;
    char* x;
    unsigned long long* dr;
    const size_t bytes = (size_t)rows * T * TILE;
    CHECK(hipMalloc(&x, bytes + (1 << 20)));
    CHECK(hipMalloc(&dr, blocks * 8));
    CHECK(hipMemset(x, 1, bytes));
    printf("%d CUs, %d rows x %d tiles x 48 KiB = %.0f MB\n", blocks, rows, T, bytes / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        run(x, blocks, rows, T, 256, 0, dr, "A row-major (a crop's keys contiguous)");
        run(x, blocks, rows, T, 256, 1, dr, "B tile-interleaved, groups of 256 rows");
        run(x, blocks, rows, T, 64, 1, dr, "B tile-interleaved, groups of 64 rows");
        run(x, blocks, rows, T, 8, 1, dr, "B tile-interleaved, groups of 8 rows");
    }
    return 0;
}
