// Probe: lane <-> element mapping of ds_read_b64_tr_b8 and of the fp8 16x16x32 MFMA operands on gfx950.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe tools/probe/tr_b8_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;

// LDS image: 64 rows x 64 bytes, byte (r, c) = unique id r*64 + c (mod 256 is ambiguous, so two runs: low and high byte)
__global__ void probe_tr(int stride, int mode, unsigned long long* out, int hi) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 256];
    for (int i = threadIdx.x; i < 64 * 256; i += 64) {
        const int r = i / stride, c = i % stride;
        const int id = r * 64 + c;          // valid for c < 64
        lds[i] = (unsigned char)(hi ? (id >> 8) : (id & 255));
    }
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 4, l = lane & 15;
    unsigned addr;
    if (mode == 0) addr = (grp * 8 + (l >> 1)) * stride + (l & 1) * 8;        // guess A: lane 2q+p -> row q, cols 8p..8p+7
    else if (mode == 1) addr = (grp * 8 + (l & 7)) * stride + (l >> 3) * 8;   // guess B: lane q + 8p -> row q, cols 8p..
    else addr = (grp * 8) * stride + l * 8;                                   // raw: 16 lanes x 8 consecutive bytes of one row
    addr += (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane] = v;
}

// fp8 MFMA operand map: A[i][k] = (i == ai && k == ak), B[k][j] = (k == ak && j == bj) -> D[ai][bj] = 1
__global__ void probe_mfma(float* out) {
    const int lane = threadIdx.x;
    // A operand: lane holds 8 fp8; put 1.0 (0x38 in e4m3) at lane la, byte ja; B: lane lb, byte jb
    for (int la = 0; la < 64; la += 21)
        for (int ja = 0; ja < 8; ja += 3) {
            // find (row, k) of A element (la, ja): use B = all ones in every k -> D[row][*] = 1
            unsigned long long a = 0, b = 0x3838383838383838ull;
            if (lane == la) a = 0x38ull << (8 * ja);
            f32x4 c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
            // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
            for (int r = 0; r < 4; ++r)
                if (c[r] != 0.f && (lane & 15) == 0) out[(la / 21) * 3 + ja / 3] = (float)(4 * (lane >> 4) + r);   // row index of A element
        }
}

// k index of A element (lane la, byte ja): B one-hot in k over all lanes/bytes -> match
__global__ void probe_mfma_k(int la, int ja, float* out) {
    const int lane = threadIdx.x;
    for (int lb = 0; lb < 64; ++lb)
        for (int jb = 0; jb < 8; ++jb) {
            unsigned long long a = 0, b = 0;
            if (lane == la) a = 0x38ull << (8 * ja);
            if (lane == lb) b = 0x38ull << (8 * jb);
            f32x4 c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
            float s = c[0] + c[1] + c[2] + c[3];
            // any lane nonzero?
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0 && s != 0.f) { out[0] = (float)lb; out[1] = (float)jb; }
            if (s != 0.f) { for (int r = 0; r < 4; ++r) if (c[r] != 0.f) { out[2] = (float)(4 * (lane >> 4) + r); out[3] = (float)(lane & 15); } }
        }
}

int main() {
    unsigned long long *d, h[64], h2[64];
    hipMalloc(&d, 64 * 8);
    for (int stride : {64, 784}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, stride, mode, d, 0); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, stride, mode, d, 1); hipMemcpy(h2, d, 512, hipMemcpyDeviceToHost);
            printf("== stride %d mode %d: lane -> 8 x (row,col) of the bytes it received\n", stride, mode);
            for (int lane = 0; lane < 64; ++lane) {
                if (lane >= 18 && lane < 32) continue;
                if (lane >= 34) continue;
                printf(" lane %2d:", lane);
                for (int j = 0; j < 8; ++j) {
                    int id = (int)((h[lane] >> (8 * j)) & 255) | ((int)((h2[lane] >> (8 * j)) & 255) << 8);
                    printf(" (%d,%d)", id / 64, id % 64);
                }
                printf("\n");
            }
        }
    }
    float *df, hf[16];
    hipMalloc(&df, 64);
    for (int la : {0, 5, 17, 37, 63})
        for (int ja : {0, 3, 7}) {
            hipMemset(df, 0, 64);
            hipLaunchKernelGGL(probe_mfma_k, dim3(1), dim3(64), 0, 0, la, ja, df);
            hipMemcpy(hf, df, 16, hipMemcpyDeviceToHost);
            printf("A(lane %2d, byte %d) pairs with B(lane %2.0f, byte %1.0f): D row %2.0f col %2.0f\n", la, ja, hf[0], hf[1], hf[2], hf[3]);
        }
    return 0;
}
