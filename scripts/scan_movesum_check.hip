// movesum_scan_kernel against the sequential recurrence (= oracle/movesum.c), on adversarial arrays:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I boss-runs_amd/csrc -I scripts scripts/scan_movesum_check.hip -o scripts/scan_movesum_check.bin
// Prints, per case, the number of (window, strand, bin) values that differ and the kernel's stretch statistics.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "engine.hpp"
#include "kernels.hip.inc"
#include "experiments/movesum_scan.hip.inc"
using namespace bossx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void serial(const std::vector<double> &a, int64_t w, int dir, std::vector<double> &y) {
    const int64_t n = int64_t(a.size());
    y.resize(size_t(n));
    double s = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        const double cur = dir ? a[size_t(j)] : a[size_t(n - 1 - j)];
        double d = cur;
        if (j >= w) d = cur - (dir ? a[size_t(j - w)] : a[size_t(n - 1 - (j - w))]);
        s = s + d;
        y[size_t(dir ? j : n - 1 - j)] = s;
    }
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 700000;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    std::mt19937_64 rng(12345);
    std::gamma_distribution<double> gam(2.0, 3.0);
    std::normal_distribution<double> nor(8.0, 0.8);
    const int32_t wins[BOSSX_NWIN] = {4, 7, 12, 21, 33, 47, 64, 90, 128, 301, 999};
    int64_t off[2] = {0, n};
    uint8_t local1 = 1;
    int64_t *d_off; uint8_t *d_local; double *d_a, *d_S; unsigned long long *d_stats; Ctrl *d_ctrl;
    CK(hipMalloc(&d_off, sizeof(off))); CK(hipMemcpy(d_off, off, sizeof(off), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_local, 1)); CK(hipMemcpy(d_local, &local1, 1, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_a, size_t(n) * 8)); CK(hipMalloc(&d_S, size_t(n) * 8 * 2 * BOSSX_NWIN));
    CK(hipMalloc(&d_stats, 64)); CK(hipMalloc(&d_ctrl, sizeof(Ctrl))); CK(hipMemset(d_ctrl, 0, sizeof(Ctrl)));
    const char *names[] = {"gamma", "hover (window sums next to powers of two)", "few mantissa bits (ties)", "zero runs", "tiny + huge", "signed"};
    int bad_total = 0;
    for (int kind = 0; kind < 6; ++kind) {
        std::vector<double> a(size_t(n), 0.0);
        for (int64_t i = 0; i < n; ++i) {
            double v = 0;
            switch (kind) {
            case 0: v = gam(rng); break;
            case 1: v = std::fabs(nor(rng)); break;
            case 2: v = double(rng() & 1023) / 16.0; break;
            case 3: v = ((i / 5000) % 3 == 1) ? 0.0 : gam(rng); break;
            case 4: v = gam(rng) * ((i / 20000) % 2 ? 1e-12 : 1e9); break;
            default: v = nor(rng) - 8.0; break;
            }
            a[size_t(i)] = v;
        }
        CK(hipMemcpy(d_a, a.data(), size_t(n) * 8, hipMemcpyHostToDevice));
        CK(hipMemset(d_stats, 0, 64));
        ChainParams P{};
        P.ds = d_a; P.benefit = nullptr; P.ctrl = d_ctrl; P.B = n; P.nb = 1; P.gate = 0; P.max_limit = n;
        P.ct.bin_off = d_off; P.ct.local = d_local; P.ct.n = 1;
        for (int k = 0; k < BOSSX_NWIN; ++k) P.w[k] = wins[k];
        if (getenv("SCAN_DEBUG_AT")) P.never_ready = atoi(getenv("SCAN_DEBUG_AT"));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(d_stats, 0, 64));
            CK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL((movesum_scan_kernel<kScanThreads, kScanG>), dim3(2 * BOSSX_NWIN), dim3(kScanThreads), 0, nullptr, P, d_S, d_stats);
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        std::vector<double> S(size_t(n) * 2 * BOSSX_NWIN);
        CK(hipMemcpy(S.data(), d_S, S.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long st[4]; CK(hipMemcpy(st, d_stats, 32, hipMemcpyDeviceToHost));
        long long bad = 0; long long first = -1; int fk = -1, fd = -1;
        std::vector<double> y;
        for (int dir = 0; dir < 2; ++dir)
            for (int k = 0; k < BOSSX_NWIN; ++k) {
                serial(a, wins[k], dir, y);
                const double *g = S.data() + (size_t(dir) * BOSSX_NWIN + size_t(k)) * size_t(n);
                for (int64_t j = 0; j < n; ++j)
                    if (memcmp(&g[j], &y[size_t(j)], 8)) { if (first < 0) { first = j; fk = k; fd = dir; } ++bad; }
            }
        printf("%-46s n=%lld: %lld values differ%s | %.3f ms | stretches %llu, ended early %llu, plain rounds %llu\n", names[kind], (long long)n, bad,
               bad ? "  <-- MISMATCH" : "", best, st[0], st[1], st[2]);
        if (bad) {
            bad_total += 1;
            // the first walk position that differs, per (window, strand)
            for (int dir = 0; dir < 2; ++dir)
                for (int k = 0; k < BOSSX_NWIN; ++k) {
                    serial(a, wins[k], dir, y);
                    const double *g = S.data() + (size_t(dir) * BOSSX_NWIN + size_t(k)) * size_t(n);
                    for (int64_t j = 0; j < n; ++j) {
                        const int64_t pos = dir ? j : n - 1 - j;
                        if (memcmp(&g[pos], &y[size_t(pos)], 8)) {
                            const int64_t pm = dir ? j - 1 : n - j;
                            printf("   window %d strand %d: first differs at walk %lld (mod 8192: %lld): got %.17g want %.17g | before: %.17g\n", wins[k], dir, (long long)j,
                                   (long long)(j % 8192), g[pos], y[size_t(pos)], j ? y[size_t(pm)] : 0.0);
                            break;
                        }
                    }
                }
        }
    }
    return bad_total ? 1 : 0;
}
