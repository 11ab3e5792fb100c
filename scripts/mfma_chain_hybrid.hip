// The FP64 matrix-core recurrence with operands from LDS (ds_read_b64, cheap to issue) and the carries
// leaving through GLOBAL stores (the vector-memory path instead of the LDS pipe the other waves load):
// G dependent v_mfma_f64_4x4x4_4b_f64 back to back, then the group's LDS reads and global stores.
//   WMODE 0: every lane stores (512 contiguous bytes)   1: lanes with lo2 == 0 of chains < 11 store, one row per chain
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/mfma_chain_hybrid.hip -o scripts/mfma_chain_hybrid.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int G, int WEVERY, int WMODE>
__global__ __launch_bounds__(64) void k(double *go, double *out, long long *t, int iters) {
    __shared__ __align__(16) double s_a[64 * 80];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 80; i += 64) s_a[i] = 1e-9 * (i + 1);
    __syncthreads();
    const int hi4 = lane >> 4, blk = (lane >> 2) & 3, lo2 = lane & 3;
    const int chain = 4 * blk + hi4;
    const bool stores = WMODE == 0 || (lo2 == 0 && chain < 11);
    double *gp = WMODE == 0 ? go + lane : go + chain * 4096;      // WMODE 1: row per chain, column = step
    double acc = 1.0 + lane * 1e-3;
    double av[2][G];
#pragma unroll
    for (int u = 0; u < G; ++u) { av[0][u] = s_a[u * 64 + lane]; av[1][u] = s_a[(u + G) * 64 + lane]; }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 64 / G; ++g) {
            double r[G];
#pragma unroll
            for (int u = 0; u < G; ++u) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g & 1][u], 1.0, acc, 0, 0, 0); r[u] = acc; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u) av[g & 1][u] = s_a[((g * G + u) & 63) * 64 + lane];
            if (WEVERY > 0) {
#pragma unroll
                for (int u = WEVERY - 1; u < G; u += WEVERY) {
                    const int step = g * G + u;
                    if (stores) { if (WMODE == 0) gp[step * 64] = r[u]; else gp[step] = r[u]; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = acc;
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

template <int G, int WEVERY, int WMODE>
void run(const char *name, double *go, double *out, long long *t) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<G, WEVERY, WMODE>), dim3(1), dim3(64), 0, 0, go, out, t, iters); hipDeviceSynchronize(); }
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double steps = double(iters) * 64;
    printf("%-58s %.2f cycles/step  %.3f ns/bin\n", name, h[0] / steps, h[1] * 10.0 / steps / 4);
}

int main() {
    double *go, *out; long long *t;
    hipMalloc(&go, 16 * 4096 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&t, 16);
    run<1, 2, 0>("G=1 LDS reads, no stores (WEVERY>G)", go, out, t);
    run<2, 2, 0>("G=2 LDS reads, global store every 2nd, all lanes", go, out, t);
    run<2, 2, 1>("G=2 LDS reads, global store every 2nd, 11 lanes", go, out, t);
    run<4, 2, 0>("G=4 LDS reads, global store every 2nd, all lanes", go, out, t);
    run<4, 2, 1>("G=4 LDS reads, global store every 2nd, 11 lanes", go, out, t);
    run<4, 4, 0>("G=4 LDS reads, global store every 4th, all lanes", go, out, t);
    run<4, 4, 1>("G=4 LDS reads, global store every 4th, 11 lanes", go, out, t);
    run<8, 2, 1>("G=8 LDS reads, global store every 2nd, 11 lanes", go, out, t);
    run<8, 4, 1>("G=8 LDS reads, global store every 4th, 11 lanes", go, out, t);
    run<8, 8, 1>("G=8 LDS reads, global store every 8th, 11 lanes", go, out, t);
    run<8, 8, 0>("G=8 LDS reads, global store every 8th, all lanes", go, out, t);
    run<4, 1, 1>("G=4 LDS reads, global store every step, 11 lanes", go, out, t);
    return 0;
}
