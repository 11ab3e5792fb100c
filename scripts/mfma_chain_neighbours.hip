// Does what the OTHER waves of the block do change the pace of a dependent MFMA chain on wave 0?
// wave 0: groups of 8 dependent v_mfma_f64_4x4x4_4b_f64 (no LDS traffic of its own); waves 1..15 (only
// those not on wave 0's SIMD do anything) spin in one of several ways until wave 0 is done.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/mfma_chain_neighbours.hip -o scripts/mfma_chain_neighbours.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0 neighbours exit; 1 s_sleep only; 2 s_sleep + LDS poll; 3 s_sleep + s_memrealtime; 4 LDS poll without sleep; 5 pure SALU spin; 6 VALU spin
__global__ __launch_bounds__(1024) void k(double *out, long long *t, int iters) {
    __shared__ unsigned s_flag[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) s_flag[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 0) {
        double acc = 1.0 + lane * 1e-3;
        const double a = 1e-9 * (lane + 1);
        const long long c0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const long long c1 = clock64();
        out[lane] = acc;
        if (lane == 0) { t[0] = c1 - c0; __hip_atomic_store(&s_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    } else if ((wave & 3) != 0 && MODE != 0) {
        unsigned long long junk = 0;
        double v = lane;
        while (true) {
            if (MODE == 1 || MODE == 2 || MODE == 3) __builtin_amdgcn_s_sleep(2);
            unsigned f = 0;
            if (MODE == 2 || MODE == 4) f = __hip_atomic_load(&s_flag[lane & 31], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) junk += wall_clock64();
            if (MODE == 6) { v = v * 1.0000001 + 0.5; }
            if (MODE == 2 || MODE == 4) { if (__builtin_amdgcn_readlane(f, 0)) break; }
            else { junk += 1; if ((junk & 0xff) == 0 && __hip_atomic_load(&s_flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break; }
        }
        if (junk == 12345 || v == 1.5) out[threadIdx.x & 63] = 1.0;
    }
}

template <int MODE>
void run(const char *name, double *out, long long *t) {
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(1024), 0, 0, out, t, iters); hipDeviceSynchronize(); }
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-46s %.2f cycles per dependent op\n", name, h[0] / (double(iters) * 64));
}

int main() {
    double *out; long long *t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 16);
    run<0>("neighbours exit", out, t);
    run<1>("neighbours: s_sleep 2 loop", out, t);
    run<2>("neighbours: s_sleep 2 + LDS poll", out, t);
    run<3>("neighbours: s_sleep 2 + s_memrealtime", out, t);
    run<4>("neighbours: LDS poll, no sleep", out, t);
    run<5>("neighbours: scalar spin", out, t);
    run<6>("neighbours: FP64 VALU spin", out, t);
    return 0;
}
