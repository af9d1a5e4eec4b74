// Host <-> device transfer options for the drop-in host-array calls (2 GiB):
//   pageable hipMemcpy, hipHostMalloc'ed buffers (allocation cost + DMA rate), hipHostRegister of an existing pageable
//   buffer (registration cost), and a 4-thread staged copy through small pinned chunks.
// build: hipcc -O3 --offload-arch=gfx950 tools/pcibench.hip -o tools/pcibench.bin -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t B = (size_t)2 << 30;
    char* d;
    CK(hipMalloc(&d, B));
    char* pg = (char*)malloc(B);
    memset(pg, 1, B);                                   // touched pageable memory (like a NumPy array that holds data)
    double t = now();
    CK(hipMemcpy(d, pg, B, hipMemcpyHostToDevice));
    printf("pageable H2D 2 GiB            %7.1f ms  %5.1f GB/s\n", (now() - t) * 1e3, B / (now() - t) / 1e9);
    t = now();
    CK(hipMemcpy(pg, d, B, hipMemcpyDeviceToHost));
    printf("pageable D2H 2 GiB            %7.1f ms  %5.1f GB/s\n", (now() - t) * 1e3, B / (now() - t) / 1e9);
    char* fresh = (char*)malloc(B);                      // untouched destination, like np.empty
    t = now();
    CK(hipMemcpy(fresh, d, B, hipMemcpyDeviceToHost));
    printf("pageable D2H into untouched   %7.1f ms  %5.1f GB/s\n", (now() - t) * 1e3, B / (now() - t) / 1e9);
    t = now();
    char* pin;
    CK(hipHostMalloc(&pin, B, hipHostMallocDefault));
    const double t_alloc = now() - t;
    t = now();
    CK(hipMemcpy(pin, d, B, hipMemcpyDeviceToHost));
    const double t_dma = now() - t;
    printf("hipHostMalloc 2 GiB           %7.1f ms; D2H into it %7.1f ms  %5.1f GB/s; together %7.1f ms\n", t_alloc * 1e3, t_dma * 1e3, B / t_dma / 1e9, (t_alloc + t_dma) * 1e3);
    t = now();
    CK(hipMemcpy(d, pin, B, hipMemcpyHostToDevice));
    printf("pinned H2D 2 GiB              %7.1f ms  %5.1f GB/s\n", (now() - t) * 1e3, B / (now() - t) / 1e9);
    t = now();
    CK(hipHostFree(pin));
    printf("hipHostFree                   %7.1f ms\n", (now() - t) * 1e3);
    t = now();
    CK(hipHostRegister(pg, B, hipHostRegisterDefault));
    const double t_reg = now() - t;
    t = now();
    CK(hipMemcpy(d, pg, B, hipMemcpyHostToDevice));
    const double t_h2d = now() - t;
    t = now();
    CK(hipHostUnregister(pg));
    printf("hipHostRegister 2 GiB         %7.1f ms; H2D from it %7.1f ms  %5.1f GB/s; unregister %7.1f ms\n", t_reg * 1e3, t_h2d * 1e3, B / t_h2d / 1e9, (now() - t) * 1e3);
    // staged: T threads, each with its own 32 MiB pinned chunk and stream
    for (int T : {2, 4, 8}) {
        const size_t CH = (size_t)32 << 20;
        std::vector<char*> pins(T);
        std::vector<hipStream_t> st(T);
        for (int i = 0; i < T; ++i) { CK(hipHostMalloc(&pins[i], CH, hipHostMallocDefault)); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
        for (int dir = 0; dir < 2; ++dir) {
            t = now();
            std::vector<std::thread> th;
            for (int i = 0; i < T; ++i)
                th.emplace_back([&, i] {
                    hipSetDevice(0);
                    for (size_t c = i; c * CH < B; c += T) {
                        if (dir == 0) {
                            hipStreamSynchronize(st[i]);
                            memcpy(pins[i], pg + c * CH, CH);
                            hipMemcpyAsync(d + c * CH, pins[i], CH, hipMemcpyHostToDevice, st[i]);
                        } else {
                            hipMemcpyAsync(pins[i], d + c * CH, CH, hipMemcpyDeviceToHost, st[i]);
                            hipStreamSynchronize(st[i]);
                            memcpy(fresh + c * CH, pins[i], CH);
                        }
                    }
                    hipStreamSynchronize(st[i]);
                });
            for (auto& x : th) x.join();
            printf("staged %s, %d threads x 32 MiB chunks   %7.1f ms  %5.1f GB/s\n", dir == 0 ? "H2D" : "D2H", T, (now() - t) * 1e3, B / (now() - t) / 1e9);
        }
        for (int i = 0; i < T; ++i) { hipHostFree(pins[i]); hipStreamDestroy(st[i]); }
    }
    return 0;
}
