// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layout, accumulation order / rounding,
// and dependent-issue latency.  Used to decide whether the FP64 matrix core can run the
// Bottleneck move_sum recurrence (4 chain steps per instruction) bit-exactly.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scripts/mfma_f64_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

__global__ void layout_kernel(double *out) {
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0;
    const double b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[blockIdx.x * 64 + lane] = d;
}

// generic: per-lane a, b, c in; d out
__global__ void mfma_kernel(const double *a, const double *b, const double *c, double *d, int n) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n) return;
    d[t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + lane], b[t * 64 + lane], c[t * 64 + lane], 0, 0, 0);
}

__global__ void latency_kernel(double *out, long long *cyc, int iters) {
    const int lane = threadIdx.x;
    double acc = 1.0 + lane * 1e-3;
    const double a = 1.0, b = 1e-9 * (lane + 1);
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    long long t1 = clock64();
    double acc2 = 1.0 + lane * 1e-3;
    long long t2 = clock64();
    for (int i = 0; i < iters; ++i) acc2 = acc2 + b;
    long long t3 = clock64();
    // mfma + quad broadcast (what the chain needs)
    double acc3 = 1.0 + lane * 1e-3;
    long long t4 = clock64();
    for (int i = 0; i < iters; ++i) {
        acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);
        int lo = __builtin_amdgcn_mov_dpp(__double2loint(acc3), 0xFF, 0xF, 0xF, true);   // quad_perm [3,3,3,3]
        int hi = __builtin_amdgcn_mov_dpp(__double2hiint(acc3), 0xFF, 0xF, 0xF, true);
        acc3 = __hiloint2double(hi, lo);
    }
    long long t5 = clock64();
    out[lane] = acc + acc2 + acc3;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; cyc[2] = t5 - t4; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
    // ---- layout ----
    std::vector<double> out(4096 * 64);
    double *d_out; CK(hipMalloc(&d_out, out.size() * 8));
    hipLaunchKernelGGL(layout_kernel, dim3(4096), dim3(64), 0, 0, d_out);
    CK(hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost));
    // for each A lane la: which B lanes pair with it (nonzero output) and where the output lands
    int pairs = 0;
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (out[(la * 64 + lb) * 64 + l] != 0.0) { if (la < 8 || la % 16 == 0) printf(" (B%d->D%d)", lb, l); ++pairs; }
        printf("\n");
    }
    printf("total nonzero (la,lb,d) triples: %d (expect 4 blocks*4*4*4 = 256)\n", pairs);
    // derive maps: for each (la, lb) pair with output at ld: la=(blk,i,k) lb=(blk,k,j) ld=(blk,i,j)
    // Print D lane for A lanes 0..15 x B lanes 0..15 as a table
    printf("D lane table for block 0 (rows A lane 0..15, cols B lane 0..15; -1 = no product):\n");
    for (int la = 0; la < 16; ++la) {
        for (int lb = 0; lb < 16; ++lb) {
            int ld = -1;
            for (int l = 0; l < 64; ++l) if (out[(la * 64 + lb) * 64 + l] != 0.0) ld = l;
            printf("%3d", ld);
        }
        printf("\n");
    }
    // ---- latency ----
    long long *d_cyc; CK(hipMalloc(&d_cyc, 64));
    hipLaunchKernelGGL(latency_kernel, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 4096);
    long long cyc[3]; CK(hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
    printf("dependent mfma_f64_4x4x4: %.1f cycles/instr; dependent v_add_f64: %.1f cycles; mfma + 2 dpp: %.1f\n",
           cyc[0] / 4096.0, cyc[1] / 4096.0, cyc[2] / 4096.0);
    // dump raw layout for offline analysis
    FILE *f = fopen("gpurun_out/mfma_layout.txt", "w");
    if (f) {
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l)
            if (out[(la * 64 + lb) * 64 + l] != 0.0) fprintf(f, "%d %d %d\n", la, lb, l);
        fclose(f);
    }
    return 0;
}
