// lbvh_sort_test.hip — the device builder's own radix sort and scan (msk_lbvh.hip) against std::sort / a host prefix sum:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I misaki-render_amd/csrc -o gpurun_scratch/lbvh_sort_test tools/micro/lbvh_sort_test.hip
// (tests/test_lbvh_sort.py builds and runs it on the GPU box)
#include "../../misaki-render_amd/csrc/msk_lbvh.hip"
#include <cstdio>
#include <random>
#include <vector>
using namespace msklbvh;
int main() {
    std::mt19937_64 rng(7);
    int bad = 0;
    for (uint32_t n : {1u, 2u, 63u, 64u, 65u, 2047u, 2048u, 2049u, 70000u, 1000003u, 5000000u}) {
        for (int mode = 0; mode < 3; ++mode) {          // random codes / few distinct codes (long runs of equal digits) / all equal
            std::vector<unsigned long long> k(n);
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t code = mode == 0 ? (uint32_t) (rng() & 0x3fffffffu) : mode == 1 ? (uint32_t) ((rng() % 5u) * 0x01010101u) & 0x3fffffffu : 0x2aaaaaaau;
                k[i] = ((unsigned long long) code << 32) | i;
            }
            std::vector<unsigned long long> want = k;
            std::sort(want.begin(), want.end());
            const uint32_t n_tiles = (n + LB_RS_TILE - 1) / LB_RS_TILE;
            const size_t n_hist = (size_t) 256 * n_tiles, tmp_words = scan_tmp_words(std::max<size_t>(n_hist, n));
            unsigned long long *k0, *k1; uint32_t *hist, *hoff, *tmp;
            if (hipMalloc(&k0, (size_t) n * 8) || hipMalloc(&k1, (size_t) n * 8) || hipMalloc(&hist, n_hist * 4) || hipMalloc(&hoff, n_hist * 4) || hipMalloc(&tmp, tmp_words * 4)) return 2;
            (void) hipMemcpy(k0, k.data(), (size_t) n * 8, hipMemcpyHostToDevice);
            if (sort_by_morton(nullptr, k0, k1, n, hist, hoff, tmp) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("n %u: HIP error\n", n); return 2; }
            std::vector<unsigned long long> got(n);
            (void) hipMemcpy(got.data(), k1, (size_t) n * 8, hipMemcpyDeviceToHost);
            const bool ok = got == want;
            // the scan on its own: random 0 / 1 flags and larger values
            std::vector<uint32_t> f(n), ex(n), sc(n);
            for (uint32_t i = 0; i < n; ++i) f[i] = mode == 0 ? (uint32_t) (rng() & 1u) : mode == 1 ? (uint32_t) (rng() % 1000u) : 1u;
            uint32_t run = 0; for (uint32_t i = 0; i < n; ++i) { ex[i] = run; run += f[i]; }
            uint32_t *din = (uint32_t *) k0, *dout = (uint32_t *) k1;
            (void) hipMemcpy(din, f.data(), (size_t) n * 4, hipMemcpyHostToDevice);
            exclusive_scan(nullptr, din, dout, n, tmp);
            (void) hipDeviceSynchronize();
            (void) hipMemcpy(sc.data(), dout, (size_t) n * 4, hipMemcpyDeviceToHost);
            const bool ok2 = sc == ex;
            printf("n %8u mode %d: sort %s, scan %s\n", n, mode, ok ? "ok" : "WRONG", ok2 ? "ok" : "WRONG");
            bad += !ok + !ok2;
            (void) hipFree(k0); (void) hipFree(k1); (void) hipFree(hist); (void) hipFree(hoff); (void) hipFree(tmp);
        }
    }
    printf(bad ? "FAILED (%d)\n" : "all ok\n", bad);
    return bad ? 1 : 0;
}
