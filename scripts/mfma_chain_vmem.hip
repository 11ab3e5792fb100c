// Feeding the FP64 matrix-core recurrence WITHOUT the LDS pipe: G dependent v_mfma_f64_4x4x4_4b_f64 back to
// back (accumulate forwarding), then one burst of GLOBAL loads (agent scope: served by the L2) for the
// operands of the group R-1 groups ahead and, every WEVERY-th step, a global store of the carry.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/mfma_chain_vmem.hip -o scripts/mfma_chain_vmem.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int G, int R, int WEVERY, bool NT>
__global__ __launch_bounds__(64) void k(const double *ga, double *go, double *out, long long *t, int iters) {
    const int lane = threadIdx.x;
    constexpr int NG = 64 / G;
    static_assert(NG % R == 0, "ring of register groups");
    double acc = 1.0 + lane * 1e-3;
    double av[R][G];
    auto ld = [&](int step) {
        const double *p = ga + (step & 63) * 64 + lane;
        return NT ? __builtin_nontemporal_load(p) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
#pragma unroll
    for (int r = 0; r < R - 1; ++r)
#pragma unroll
        for (int u = 0; u < G; ++u) av[r][u] = ld(r * G + u);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double r[G];
#pragma unroll
            for (int u = 0; u < G; ++u) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[g % R][u], 1.0, acc, 0, 0, 0); r[u] = acc; }
            __builtin_amdgcn_sched_barrier(0);
            // burst: operands of group g + R - 1 into the register group this one has just left ... wait: that is group (g + R - 1) % R
#pragma unroll
            for (int u = 0; u < G; ++u) av[(g + R - 1) % R][u] = ld((g + R - 1) * G + u);
            if (WEVERY > 0) {
#pragma unroll
                for (int u = WEVERY - 1; u < G; u += WEVERY) go[((g * G + u) & 63) * 64 + lane] = r[u];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = acc;
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

template <int G, int R, int WEVERY, bool NT>
void run(const char *name, const double *ga, double *go, double *out, long long *t) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<G, R, WEVERY, NT>), dim3(1), dim3(64), 0, 0, ga, go, out, t, iters); hipDeviceSynchronize(); }
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double steps = double(iters) * 64;
    printf("%-52s %.2f cycles/step  %.3f ns/bin\n", name, h[0] / steps, h[1] * 10.0 / steps / 4);
}

int main() {
    double *ga, *go, *out; long long *t;
    hipMalloc(&ga, 64 * 64 * 8); hipMalloc(&go, 64 * 64 * 8); hipMalloc(&out, 64 * 8); hipMalloc(&t, 16);
    double h[64 * 64];
    for (int i = 0; i < 64 * 64; ++i) h[i] = 1e-9 * (i + 1);
    hipMemcpy(ga, h, sizeof(h), hipMemcpyHostToDevice);
    run<8, 2, 0, false>("G=8 ring 2, agent loads, no stores", ga, go, out, t);
    run<8, 4, 0, false>("G=8 ring 4, agent loads, no stores", ga, go, out, t);
    run<8, 8, 0, false>("G=8 ring 8, agent loads, no stores", ga, go, out, t);
    run<8, 8, 0, true>("G=8 ring 8, nt loads, no stores", ga, go, out, t);
    run<8, 8, 8, false>("G=8 ring 8, agent loads, store every 8th", ga, go, out, t);
    run<8, 8, 4, false>("G=8 ring 8, agent loads, store every 4th", ga, go, out, t);
    run<8, 8, 2, false>("G=8 ring 8, agent loads, store every 2nd", ga, go, out, t);
    run<4, 8, 0, false>("G=4 ring 8, agent loads, no stores", ga, go, out, t);
    run<4, 16, 4, false>("G=4 ring 16, agent loads, store every 4th", ga, go, out, t);
    run<4, 16, 2, false>("G=4 ring 16, agent loads, store every 2nd", ga, go, out, t);
    run<2, 16, 2, false>("G=2 ring 16, agent loads, store every 2nd", ga, go, out, t);
    run<1, 32, 1, false>("G=1 ring 32, agent loads, store every step", ga, go, out, t);
    run<1, 32, 0, false>("G=1 ring 32, agent loads, no stores", ga, go, out, t);
    run<16, 4, 8, false>("G=16 ring 4, agent loads, store every 8th", ga, go, out, t);
    run<16, 4, 16, false>("G=16 ring 4, agent loads, store every 16th", ga, go, out, t);
    return 0;
}
