// How to feed the FP64 matrix-core recurrence: G dependent v_mfma_f64_4x4x4_4b_f64 back to back
// (accumulate forwarding), then one burst of the group's LDS traffic.  Variants of the burst:
//   R = operand reads per group (b64: G reads; b128: G/2 reads), W = carry stores per group.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/mfma_chain_groups.hip -o scripts/mfma_chain_groups.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int G, int RMODE, int WEVERY, int WLANES = 64>   // WLANES: lanes that store (exec-masked);  RMODE 0: b64 reads, 1: b128 reads (two steps each); WEVERY: store every n-th step's carry (0 = none)
__global__ __launch_bounds__(64) void k(double *out, long long *t, int iters) {
    __shared__ __align__(16) double s_a[64 * 80], s_o[64 * 80];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 80; i += 64) { s_a[i] = 1e-9 * (i + 1); s_o[i] = 0.0; }
    __syncthreads();
    double acc = 1.0 + lane * 1e-3;
    double av[2][G];
#pragma unroll
    for (int u = 0; u < G; ++u) { av[0][u] = s_a[u * 64 + lane]; av[1][u] = s_a[(u + G) * 64 + lane]; }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 64 / G; ++g) {
            double r[G];
            // the group's dependent chain, nothing in between
#pragma unroll
            for (int u = 0; u < G; ++u) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g & 1][u], 1.0, acc, 0, 0, 0); r[u] = acc; }
            __builtin_amdgcn_sched_barrier(0);
            // burst: operands of the group after next, carries of this group
            if (RMODE == 0) {
#pragma unroll
                for (int u = 0; u < G; ++u) av[g & 1][u] = s_a[((g * G + u) & 63) * 64 + lane];
            } else {
#pragma unroll
                for (int u = 0; u < G; u += 2) {
                    const double2 v = *reinterpret_cast<const double2 *>(&s_a[((g * G + u) & 63) * 64 + 2 * (lane & 31)]);
                    av[g & 1][u] = v.x; av[g & 1][u + 1] = v.y;
                }
            }
            if (WEVERY > 0) {
#pragma unroll
                for (int u = WEVERY - 1; u < G; u += WEVERY) { if (WLANES == 64 || lane < WLANES) s_o[(g * G + u) * 64 + lane] = r[u]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = acc + s_o[lane];
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

template <int G, int RMODE, int WEVERY, int WLANES = 64>
void run(const char *name, double *out, long long *t) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<G, RMODE, WEVERY, WLANES>), dim3(1), dim3(64), 0, 0, out, t, iters); hipDeviceSynchronize(); }
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double steps = double(iters) * 64;
    printf("%-34s %.2f cycles/step  %.3f ns/bin\n", name, h[0] / steps, h[1] * 10.0 / steps / 4);
}

int main() {
    double *out; long long *t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 16);
    run<1, 0, 1>("G=1 b64 reads, store every step", out, t);
    run<2, 0, 1>("G=2 b64 reads, store every step", out, t);
    run<4, 0, 1>("G=4 b64 reads, store every step", out, t);
    run<8, 0, 1>("G=8 b64 reads, store every step", out, t);
    run<16, 0, 1>("G=16 b64 reads, store every step", out, t);
    run<8, 1, 1>("G=8 b128 reads, store every step", out, t);
    run<8, 1, 2>("G=8 b128 reads, store every 2nd", out, t);
    run<8, 1, 4>("G=8 b128 reads, store every 4th", out, t);
    run<8, 0, 4>("G=8 b64 reads, store every 4th", out, t);
    run<8, 0, 0>("G=8 b64 reads, no stores", out, t);
    run<8, 1, 0>("G=8 b128 reads, no stores", out, t);
    run<16, 1, 4>("G=16 b128 reads, store every 4th", out, t);
    run<16, 1, 1>("G=16 b128 reads, store every step", out, t);
    run<4, 0, 2>("G=4 b64 reads, store every 2nd", out, t);
    run<4, 0, 4>("G=4 b64 reads, store every 4th", out, t);
    run<8, 0, 8>("G=8 b64 reads, store every 8th", out, t);
    run<8, 0, 2>("G=8 b64 reads, store every 2nd", out, t);
    run<8, 0, 1, 16>("G=8 b64, every step, 16 lanes", out, t);
    run<8, 0, 2, 16>("G=8 b64, every 2nd, 16 lanes", out, t);
    run<8, 0, 1, 32>("G=8 b64, every step, 32 lanes", out, t);
    return 0;
}
