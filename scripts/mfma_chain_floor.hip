// Floor of the move_sum recurrence on the FP64 matrix core (gfx950): cycles per 4-bin step of
//   (a) a dependent chain of v_mfma_f64_4x4x4_4b_f64 alone,
//   (b) the chain kernel's pair: CARRY op on the chain + PREFIX op hanging off the previous carry,
//   (c) the pair plus one ds_read_b64 and one ds_write_b64 per step (the barrier kernel's whole step),
//   (d) ONE op per step plus the operand read and the carry store (the flow kernel's chain wave).
// Prints shader cycles (s_memtime) and wall time (s_memrealtime, 100 MHz) per step.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/mfma_chain_floor.hip -o scripts/mfma_chain_floor.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ long long wall() { return __builtin_readcyclecounter(); }

template <int MODE>
__global__ __launch_bounds__(64) void floor_kernel(double *out, long long *t, int iters) {
    __shared__ double s_a[64 * 64], s_o[64 * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 64; i += 64) { s_a[i] = 1e-9 * (i + 1); s_o[i] = 0.0; }
    __syncthreads();
    double acc = 1.0 + lane * 1e-3, keep = 0.0;
    const double bm = (lane >> 4) <= (lane & 3) ? 1.0 : 0.0;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < (MODE == 3 ? 0 : iters); ++i) {
        double av[64];
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 6; ++u) av[u] = s_a[u * 64 + lane];
        }
        double prev = 0.0;
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const double a = MODE == 2 ? av[u] : 1e-9 * (u + 1);
            const double c_in = acc;
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, c_in, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) {
                if (u + 6 < 64) av[u + 6] = s_a[(u + 6) * 64 + lane];
                if (u >= 1) s_o[(u - 1) * 64 + lane] = prev;
            }
            if (MODE >= 1) {
                const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bm, c_in, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 2) prev = d; else keep += d;
            }
        }
        if (MODE == 2) s_o[63 * 64 + lane] = prev;
    }
    if (MODE == 3) {
        acc = 1.0 + lane * 1e-3;
        for (int i = 0; i < iters; ++i) {
            double av[64];
#pragma unroll
            for (int u = 0; u < 12; ++u) av[u] = s_a[u * 64 + lane];
            double prev = 0.0;
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u], 1.0, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 12 < 64) av[u + 12] = s_a[(u + 12) * 64 + lane];
                s_o[u * 64 + lane] = prev;
                __builtin_amdgcn_sched_barrier(0);
                prev = acc;
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = acc + keep + s_o[lane];
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

int main() {
    double *out; long long *t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 16);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(floor_kernel<0>, dim3(1), dim3(64), 0, 0, out, t, iters);
            if (mode == 1) hipLaunchKernelGGL(floor_kernel<1>, dim3(1), dim3(64), 0, 0, out, t, iters);
            if (mode == 2) hipLaunchKernelGGL(floor_kernel<2>, dim3(1), dim3(64), 0, 0, out, t, iters);
            if (mode == 3) hipLaunchKernelGGL(floor_kernel<3>, dim3(1), dim3(64), 0, 0, out, t, iters);
            hipDeviceSynchronize();
        }
        long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double steps = double(iters) * 64;
        printf("mode %d: %.2f shader cycles/step, %.2f ns/step (%.3f ns/bin), clock %.2f GHz\n", mode, h[0] / steps,
               h[1] * 10.0 / steps, h[1] * 10.0 / steps / 4, double(h[0]) / (h[1] * 10.0));
    }
    return 0;
}
