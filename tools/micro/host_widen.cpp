// How fast can the host turn int32 into int64 (the other half of sending row indices over PCIe as 4 bytes)?  g++ -O2 -pthread
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
    const size_t n = (size_t)64 << 20;
    int32_t *src = (int32_t *)aligned_alloc(4096, n * 4);
    int64_t *dst = (int64_t *)aligned_alloc(4096, n * 8);
    for (size_t i = 0; i < n; ++i) src[i] = (int32_t)i;
    memset(dst, 0, n * 8);
    for (int nt : {1, 2, 4, 8, 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([=] {
                    const size_t a = n * t / nt, b = n * (t + 1) / nt;
                    for (size_t i = a; i < b; ++i) dst[i] = src[i];
                });
            for (auto &x : th) x.join();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s < best) best = s;
        }
        printf("%2d threads: %.2f ms for %zu M entries = %.1f G entries/s (%.1f GB/s read + written)\n", nt, best * 1e3, n >> 20, n / best / 1e9, n * 12 / best / 1e9);
    }
    // and a plain memcpy of the same output bytes, for scale
    char *a = (char *)aligned_alloc(4096, n * 8);
    memset(a, 1, n * 8);
    auto t0 = std::chrono::steady_clock::now();
    memcpy(dst, a, n * 8);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("memcpy of %zu MB, 1 thread: %.2f ms = %.1f GB/s\n", (n * 8) >> 20, s * 1e3, n * 8 / s / 1e9);
    return 0;
}
