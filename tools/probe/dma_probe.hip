// Probe (gfx950): how fast does a CU's LDS-DMA path (global_load_lds_dwordx4, 1 KiB per wave instruction) stream GEMM operand
// tiles, and does the shape of a request matter?  The encoder GEMM's K loop (csrc/kernels_gemm_pers.h) requests 32-deep
// K-tiles: a request covers 16 rows x 64 B - sixteen HALF cache lines.  A 64-deep K-tile makes a request 8 rows x 128 B -
// eight whole lines - for the same bytes.  One 512-thread block per CU streams A panels of its own (HBM / Infinity Cache)
// and one W slice shared by every block (L2), K = 768 (1,536-byte rows), no MFMA, nothing consumed.
//     hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/dma_probe tools/probe/dma_probe.hip && tools/probe/bin/dma_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// DEEP: 32 (64-byte rows, 16 rows per request) or 64 (128-byte rows, 8 rows per request); WAVES: waves that request (4 or 8)
template <int DEEP, int WAVES>
__global__ __launch_bounds__(512, 1) void dma_k(const char* A, const char* W, int panels, int row_bytes, int share, unsigned long long* rt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWB = DEEP * 2;                    // bytes of a tile row
    constexpr int RPR = 1024 / ROWB;                  // rows per request
    constexpr int TILE = 512 * ROWB;                  // 256 A rows + 256 W rows
    constexpr int SLOTS = 128 * 1024 / TILE;          // 4 (32-deep) or 2 (64-deep)
    constexpr int REQ = 512 / RPR;                    // requests per K-tile
    constexpr int PER = REQ / WAVES;                  // ... per requesting wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane / (ROWB / 16), lch = lane % (ROWB / 16);
    const int nk = row_bytes / ROWB;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    int g = 0;
    for (int pnl = 0; pnl < panels; ++pnl) {
        // `share` blocks that are neighbours in one XCD's block order (b, b + 8, ...) read the same panel, as the column
        // blocks of one A row-panel do in the GEMM (share = 1: every block streams its own panel from HBM)
        const int grp = (int)(blockIdx.x & 7) + 8 * (int)((blockIdx.x >> 3) / share);
        const char* a = A + ((size_t)(pnl * gridDim.x + grp) * 256) * row_bytes;
        for (int kt = 0; kt < nk; ++kt, ++g) {
            if (wave < WAVES) {
                char* slot = smem + (g % SLOTS) * TILE;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int rq = wave + WAVES * i;              // request index 0 .. REQ-1: first half A, second half W
                    const int row = (rq % (REQ / 2)) * RPR + lrow;
                    const char* src = (rq < REQ / 2 ? a : W) + (size_t)row * row_bytes + (size_t)kt * ROWB + lch * 16;
                    glds16(src, slot + rq * 1024);
                }
            }
            // keep SLOTS - 1 K-tiles in flight (whole K-tiles: PER requests each)
            if (wave < WAVES) {
                if constexpr (SLOTS == 4) {
                    if constexpr (PER == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                } else {
                    if constexpr (PER == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) rt[blockIdx.x] = r1 - r0;
}

template <int DEEP, int WAVES>
static void run(const char* A, const char* W, int blocks, int panels, int row_bytes, int share, unsigned long long* dr, const char* what) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_k<DEEP, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((dma_k<DEEP, WAVES>), dim3(blocks), dim3(512), 160 * 1024, 0, A, W, panels, row_bytes, share, dr);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> r(blocks);
    CHECK(hipMemcpy(r.data(), dr, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(r.begin(), r.end());
    const double bytes_per_block = (double)panels * 512.0 * row_bytes;
    const double us_med = r[blocks / 2] / 100.0;       // s_memrealtime: 100 MHz
    printf("share %2d  %-44s: launch %8.1f us  -> %6.2f TB/s chip, %6.1f GB/s per CU (median block %8.1f us)\n", share, what, ms * 1e3,
           bytes_per_block * blocks / (ms * 1e-3) / 1e12, bytes_per_block / (us_med * 1e-6) / 1e9, us_med);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    const int row_bytes = 1536, panels = 8;
    char *A, *W;
    unsigned long long* dr;
    const size_t abytes = (size_t)panels * blocks * 256 * row_bytes;
    CHECK(hipMalloc(&A, abytes + (1 << 20)));
    CHECK(hipMalloc(&W, (size_t)256 * row_bytes + (1 << 20)));
    CHECK(hipMalloc(&dr, blocks * 8));
    CHECK(hipMemset(A, 1, abytes));
    CHECK(hipMemset(W, 2, (size_t)256 * row_bytes));
    printf("%d CUs, %d panels of 256 rows x %d B per block (%.0f MB of A in all) + a shared 256-row W slice\n", blocks, panels, row_bytes, abytes / 1e6);
    for (int share : {1, 8, 32}) {
        run<32, 4>(A, W, blocks, panels, row_bytes, share, dr, "32-deep K-tiles (16 rows x 64 B), 4 waves");
        run<32, 8>(A, W, blocks, panels, row_bytes, share, dr, "32-deep K-tiles (16 rows x 64 B), 8 waves");
        run<64, 4>(A, W, blocks, panels, row_bytes, share, dr, "64-deep K-tiles (8 rows x 128 B), 4 waves");
        run<64, 8>(A, W, blocks, panels, row_bytes, share, dr, "64-deep K-tiles (8 rows x 128 B), 8 waves");
    }
    return 0;
}
