// Does v_mfma_f64_4x4x4_4b_f64 accumulate k = 0,1,2,3 sequentially with one rounding per step,
// i.e. D = fl(fl(fl(fl(C + a0*b0) + a1*b1) + a2*b2) + a3*b3) when a_k in {0,1}?
// Layout (measured, scripts/mfma_f64_probe.hip): A lane = 16k+4blk+i, B lane = 16k+4blk+j,
// D lane = 16i+4blk+j.   chain mapping: A[i][k] = d_{k} of chain i, B[k][j] = (k <= j).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k(const double *a, const double *b, const double *c, double *d, int n) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t < n) d[t * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + lane], b[t * 64 + lane], c[t * 64 + lane], 0, 0, 0);
}

int main() {
    const int n = 200000;
    std::vector<double> A(n * 64), B(n * 64), C(n * 64), D(n * 64);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    auto rnd = [&]() {   // wide dynamic range, both signs
        double m = u(rng) + 0.5; int e = int(rng() % 40) - 20; double v = std::ldexp(m, e);
        return (rng() & 1) ? v : -v; };
    for (int t = 0; t < n; ++t)
        for (int blk = 0; blk < 4; ++blk)
            for (int x = 0; x < 4; ++x)
                for (int kk = 0; kk < 4; ++kk) {
                    // chain i = x: data d_kk ; B[k][j=x] = (kk <= x)
                    A[t * 64 + 16 * kk + 4 * blk + x] = rnd();
                    B[t * 64 + 16 * kk + 4 * blk + x] = (kk <= x) ? 1.0 : 0.0;
                }
    // C[i][j] = start value of chain i (same for all j)
    for (int t = 0; t < n; ++t)
        for (int blk = 0; blk < 4; ++blk)
            for (int i = 0; i < 4; ++i) {
                const double s = rnd() * 8;
                for (int j = 0; j < 4; ++j) C[t * 64 + 16 * i + 4 * blk + j] = s;
            }
    double *dA, *dB, *dC, *dD;
    size_t bytes = size_t(n) * 64 * 8;
    CK(hipMalloc(&dA, bytes)); CK(hipMalloc(&dB, bytes)); CK(hipMalloc(&dC, bytes)); CK(hipMalloc(&dD, bytes));
    CK(hipMemcpy(dA, A.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD, n);
    CK(hipMemcpy(D.data(), dD, bytes, hipMemcpyDeviceToHost));
    long long bad_seq = 0, bad_rev = 0, total = 0;
    for (int t = 0; t < n; ++t)
        for (int blk = 0; blk < 4; ++blk)
            for (int i = 0; i < 4; ++i) {
                double d[4];
                for (int kk = 0; kk < 4; ++kk) d[kk] = A[t * 64 + 16 * kk + 4 * blk + i];
                const double s = C[t * 64 + 16 * i + 4 * blk + 0];
                for (int j = 0; j < 4; ++j) {
                    volatile double seq = s;
                    for (int kk = 0; kk <= j; ++kk) seq = seq + d[kk];
                    volatile double rev = s;
                    for (int kk = j; kk >= 0; --kk) rev = rev + d[kk];
                    const double got = D[t * 64 + 16 * i + 4 * blk + j];
                    ++total;
                    if (memcmp(&got, (const void *)&seq, 8)) ++bad_seq;
                    if (memcmp(&got, (const void *)&rev, 8)) ++bad_rev;
                }
            }
    printf("prefix sums checked: %lld  mismatches vs k-ascending sequential chain: %lld  vs k-descending: %lld\n", total, bad_seq, bad_rev);
    return 0;
}
