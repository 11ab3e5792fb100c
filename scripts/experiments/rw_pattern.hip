// What the memory system of an MI355X gives the ACCESS PATTERN of the ingest sweep, without the sweep's arithmetic:
// persistent blocks walk a list of 22,016-byte tile records (2000 sites: five u16 counter planes + 2000 state bytes), read every
// record (coalesced 16-byte vectors, the next record in flight while this one is "used") and write back, per record,
//   c : the 4-KB plane 0 + the 2-KB state bytes in place        (what a tile that receives bases changes, more or less)
//   e : 16 KB of float64 to a second array (the entropies), eight 512-byte rows per wave
// Tiles: every tile of a 1.2-GB state (stream) or a random 23 % of them in address order (scatter: what a 4000-read batch touches).
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/rw_pattern.hip -o scripts/rw_pattern.bin && scripts/rw_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int kStride = 22016, kSites = 2000;

// SP: 13 % of the threads also write their 16-byte vector of each of planes 1..4 (a batch's errors) and 5 % of the entropy lanes are off
template <bool WC, bool WE, bool PF, bool SP = false>
__global__ __launch_bounds__(256, 3) void rw_kernel(uint8_t *state, double *ent, const uint32_t *tiles, uint32_t n, unsigned long long *sink) {
    const uint32_t tid = threadIdx.x, t = tid < 250 ? tid : 249;
    uint32_t w = uint32_t((uint64_t(n) * blockIdx.x) / gridDim.x);
    const uint32_t w_end = uint32_t((uint64_t(n) * (blockIdx.x + 1)) / gridDim.x);
    if (w >= w_end) return;
    uint4 q[5]; uint2 m;
    auto load = [&](uint32_t tile) {
        const uint8_t *rec = state + size_t(tile) * kStride;
#pragma unroll
        for (int k = 0; k < 5; ++k) q[k] = *reinterpret_cast<const uint4 *>(rec + k * 4000 + t * 16);
        m = *reinterpret_cast<const uint2 *>(rec + 20000 + t * 8);
    };
    load(tiles[w]);
    unsigned long long acc = 0;
    for (; w < w_end; ++w) {
        const uint32_t tile = tiles[w];
        uint4 c[5]; uint2 cm = m;
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] = q[k];
        if (PF && w + 1 < w_end) load(tiles[w + 1]);
#pragma unroll
        for (int k = 0; k < 5; ++k) acc += c[k].x + c[k].y + c[k].z + c[k].w;
        acc += cm.x + cm.y;
        uint8_t *rec = state + size_t(tile) * kStride;
        if (WC && tid < 250) {
            c[0].x += 1u;
            *reinterpret_cast<uint4 *>(rec + tid * 16) = c[0];
            cm.x |= 4u;
            *reinterpret_cast<uint2 *>(rec + 20000 + tid * 8) = cm;
            if (SP) {
                uint32_t h = (tile * 2654435761u) ^ (tid * 40503u); h ^= h >> 13; h *= 0x9E3779B1u; h ^= h >> 16;
#pragma unroll
                for (int k = 1; k < 5; ++k) if (((h >> (7 * k)) & 127u) < 17u) *reinterpret_cast<uint4 *>(rec + k * 4000 + tid * 16) = c[k];
            }
        }
        if (WE && tid < 250) {
            double *e = ent + size_t(tile) * kSites + tid;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!SP || ((tile * 31u + tid * 7u + j * 13u) % 20u) != 0u) e[j * 250] = double(c[j % 5].x) * 0.5;
        }
        if (!PF && w + 1 < w_end) load(tiles[w + 1]);
    }
    if (acc == 0x1234567ull) atomicAdd(sink, acc);
}

// `lists`: the launches cycle through n_lists tile lists of n tiles each (different batches: a list comes round again after the
// others have pushed 4 GB through the 256-MB Infinity Cache)
template <bool WC, bool WE, bool PF, bool SP = false>
double run(uint8_t *state, double *ent, const uint32_t *d_tiles, uint32_t n, unsigned long long *sink, int grid, int n_lists = 1) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rw_kernel<WC, WE, PF, SP>), dim3(grid), dim3(256), 0, 0, state, ent, d_tiles + size_t(i % n_lists) * n, n, sink);
    CK(hipEventRecord(a));
    const int reps = 24;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rw_kernel<WC, WE, PF, SP>), dim3(grid), dim3(256), 0, 0, state, ent, d_tiles + size_t(i % n_lists) * n, n, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return double(ms) / reps;
}

// The granularity question: records of pitch `pitch` per plane (4000: the engine's; 4096: planes on 128-byte boundaries); a thread's
// vector of planes 1..4 goes back if ANY lane of its group of G lanes drew "dirty" (13.3 % per lane and plane): G = 1 isolated 16-byte
// writes, 2 / 4 / 8 whole 32 / 64 / 128-byte stretches.
template <int G>
__global__ __launch_bounds__(256, 3) void sparse_kernel(uint8_t *state, const uint32_t *tiles, uint32_t n, unsigned long long *sink, int pitch, int stride) {
    const uint32_t tid = threadIdx.x, t = tid < 250 ? tid : 249;
    uint32_t w = uint32_t((uint64_t(n) * blockIdx.x) / gridDim.x);
    const uint32_t w_end = uint32_t((uint64_t(n) * (blockIdx.x + 1)) / gridDim.x);
    if (w >= w_end) return;
    uint4 q[5];
    auto load = [&](uint32_t tile) {
        const uint8_t *rec = state + size_t(tile) * stride;
#pragma unroll
        for (int k = 0; k < 5; ++k) q[k] = *reinterpret_cast<const uint4 *>(rec + k * pitch + t * 16);
    };
    load(tiles[w]);
    unsigned long long acc = 0;
    for (; w < w_end; ++w) {
        const uint32_t tile = tiles[w];
        uint4 c[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) c[k] = q[k];
        if (w + 1 < w_end) load(tiles[w + 1]);
#pragma unroll
        for (int k = 0; k < 5; ++k) acc += c[k].x + c[k].y + c[k].z + c[k].w;
        uint8_t *rec = state + size_t(tile) * stride;
        uint32_t h = (tile * 2654435761u) ^ (tid * 40503u); h ^= h >> 13; h *= 0x9E3779B1u; h ^= h >> 16;
        uint32_t dirty = 1u;
#pragma unroll
        for (int k = 1; k < 5; ++k) if (((h >> (7 * k)) & 127u) < 17u) dirty |= 1u << k;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) dirty |= uint32_t(__shfl_xor(int(dirty), d, 64));
        if (tid < 250) {
#pragma unroll
            for (int k = 0; k < 5; ++k) if ((dirty >> k) & 1u) { c[k].x += 1u; *reinterpret_cast<uint4 *>(rec + k * pitch + tid * 16) = c[k]; }
        }
    }
    if (acc == 0x1234567ull) atomicAdd(sink, acc);
}
template <int G>
double run_sparse(uint8_t *state, const uint32_t *d_tiles, uint32_t n, unsigned long long *sink, int grid, int n_lists, int pitch, int stride) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((sparse_kernel<G>), dim3(grid), dim3(256), 0, 0, state, d_tiles + size_t(i % n_lists) * n, n, sink, pitch, stride);
    CK(hipEventRecord(a));
    const int reps = 24;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((sparse_kernel<G>), dim3(grid), dim3(256), 0, 0, state, d_tiles + size_t(i % n_lists) * n, n, sink, pitch, stride);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return double(ms) / reps;
}

__global__ void copy_kernel(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) b[i] = a[i];
}

int main() {
    const uint32_t n_all = 55600;                       // chr20+21
    uint8_t *state; double *ent; unsigned long long *sink;
    CK(hipMalloc(&state, size_t(n_all) * kStride)); CK(hipMalloc(&ent, size_t(n_all) * kSites * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(state, 1, size_t(n_all) * kStride)); CK(hipMemset(ent, 0, size_t(n_all) * kSites * 8));
    std::vector<uint32_t> all(n_all), sc;
    for (uint32_t i = 0; i < n_all; ++i) all[i] = i;
    // touched tiles: 4000 reads of ~4 consecutive tiles each
    const int kLists = 8; const uint32_t n_sc = 12000;
    for (int l = 0; l < kLists; ++l) {
        std::mt19937 rng(5 + l); std::vector<uint8_t> hit(n_all, 0); std::vector<uint32_t> one;
        while (one.size() < n_sc) {
            one.clear();
            for (int r = 0; r < 4000; ++r) { uint32_t s = rng() % (n_all - 5); int len = 2 + rng() % 4; for (int k = 0; k < len; ++k) hit[s + k] = 1; }
            for (uint32_t i = 0; i < n_all; ++i) if (hit[i]) one.push_back(i);
        }
        one.resize(n_sc);
        sc.insert(sc.end(), one.begin(), one.end());
    }
    uint32_t *d_all, *d_sc;
    CK(hipMalloc(&d_all, n_all * 4)); CK(hipMalloc(&d_sc, sc.size() * 4));
    CK(hipMemcpy(d_all, all.data(), n_all * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    printf("tiles: all %u, scattered %u per list, %d lists\n", n_all, n_sc, kLists);
    {   // float4 copy ceiling on this box
        const size_t nv = size_t(n_all) * kStride / 16 / 2;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(state), reinterpret_cast<uint4 *>(state) + nv, nv);
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(state), reinterpret_cast<uint4 *>(state) + nv, nv);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("float4 copy of %.0f MB: %.3f ms = %.2f TB/s (read + write)\n", nv * 16 / 1e6, ms / 10, 2.0 * nv * 16 / (ms / 10 * 1e-3) / 1e12);
    }
    for (int grid : {256 * 2, 256 * 3}) {
        for (int which = 0; which < 2; ++which) {
            const uint32_t *tl = which ? d_sc : d_all; const uint32_t n = which ? n_sc : n_all;
            const int nl = which ? kLists : 1;
            const char *nm = which ? "scatter" : "stream ";
            const double rd = double(n) * kStride, wc = double(n) * 6000.0, we = double(n) * 16000.0;
            auto line = [&](const char *what, double ms, double bytes) { printf("  grid %4d %s %-22s %.4f ms  %7.1f MB  %.2f TB/s\n", grid, nm, what, ms, bytes / 1e6, bytes / (ms * 1e-3) / 1e12); };
            line("read only", run<false, false, true>(state, ent, tl, n, sink, grid, nl), rd);
            line("read + counters", run<true, false, true>(state, ent, tl, n, sink, grid, nl), rd + wc);
            line("read + counters, sparse", run<true, false, true, true>(state, ent, tl, n, sink, grid, nl), rd + wc + double(n) * 4 * 250 * 0.133 * 16);
            line("read + entropy", run<false, true, true>(state, ent, tl, n, sink, grid, nl), rd + we);
            line("read + both", run<true, true, true>(state, ent, tl, n, sink, grid, nl), rd + wc + we);
            line("read + both, sparse", run<true, true, true, true>(state, ent, tl, n, sink, grid, nl), rd + wc + double(n) * 4 * 250 * 0.133 * 16 + we * 0.95);
        }
    }
    printf("write-back granularity (plane 0 always, planes 1..4 where a lane of the group of G drew dirty), grid 768, scattered lists:\n");
    for (int lay = 0; lay < 2; ++lay) {
        const int pitch = lay ? 4096 : 4000, stride = lay ? 22528 : 22016;
        const uint32_t n = n_sc * 22016ull / stride >= n_sc ? n_sc : n_sc;      // the same tile indices (the state holds 55,600 records of the larger stride too? no: cap)
        uint8_t *st2 = state;
        if (lay) { CK(hipMalloc(&st2, size_t(n_all) * stride)); CK(hipMemset(st2, 1, size_t(n_all) * stride)); }
        auto frac = [](int g) { double p = 1.0; for (int i = 0; i < g; ++i) p *= 0.867; return 1.0 - p; };
        const double t1 = run_sparse<1>(st2, d_sc, n, sink, 768, kLists, pitch, stride), t2 = run_sparse<2>(st2, d_sc, n, sink, 768, kLists, pitch, stride),
                     t4 = run_sparse<4>(st2, d_sc, n, sink, 768, kLists, pitch, stride), t8 = run_sparse<8>(st2, d_sc, n, sink, 768, kLists, pitch, stride),
                     t64 = run_sparse<64>(st2, d_sc, n, sink, 768, kLists, pitch, stride);
        printf("  plane pitch %d: G=1 %.4f ms (%.0f B/tile in planes 1-4)  G=2 %.4f (%.0f)  G=4 %.4f (%.0f)  G=8 %.4f (%.0f)  G=64 %.4f (all)\n", pitch,
               t1, 16000 * frac(1), t2, 16000 * frac(2), t4, 16000 * frac(4), t8, 16000 * frac(8), t64);
    }
    return 0;
}
