// tlb_probe.hip -- latency of dependent scattered 8-byte loads as a function of the working set and of HOW the memory
// was allocated: one hipMalloc of the whole set, or many allocations of a few MB (what a batch of handles is).
//   hipcc --offload-arch=gfx950 -O3 -o tlb_probe tools/tlb_probe.hip && ./tlb_probe
// Each of W waves (one lane active) walks its own random cycle of NODES nodes spread over the set; reported: ns per hop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
__global__ void scat(unsigned long long *a, unsigned long long *v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *(unsigned long long *)a[i] = v[i];
}
__global__ void chase(unsigned long long **starts, int hops, unsigned long long *sink)
{
    if ((threadIdx.x & 63) != 0) return;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long *p = starts[w];
    for (int h = 0; h < hops; h++) p = (unsigned long long *)*p;
    sink[w] = (unsigned long long)p;
}
int main(int argc, char **argv)
{
    const size_t GB = 1ull << 30;
    const double sets_gb[] = {0.25, 2, 16, 64, 200};
    const int waves_list[] = {256, 4096};
    for (int mode = 0; mode < 2; mode++) { // 0: one allocation, 1: pieces of 4-32 MB
        for (double sg : sets_gb) {
            const size_t total = (size_t)(sg * GB);
            std::vector<char *> pieces;
            std::vector<size_t> psize;
            std::mt19937_64 rng(12345);
            if (mode == 0) {
                char *p = nullptr;
                if (hipMalloc(&p, total) != hipSuccess) { printf("mode %d set %.2f GB: hipMalloc failed\n", mode, sg); continue; }
                pieces.push_back(p); psize.push_back(total);
            } else {
                size_t got = 0; bool ok = true;
                while (got < total) {
                    const size_t sz = ((rng() % 8) + 1) * (4ull << 20);
                    char *p = nullptr;
                    if (hipMalloc(&p, sz) != hipSuccess) { ok = false; break; }
                    pieces.push_back(p); psize.push_back(sz); got += sz;
                }
                if (!ok) { printf("mode %d set %.2f GB: hipMalloc failed\n", mode, sg); for (auto q : pieces) hipFree(q); continue; }
            }
            for (int W : waves_list) {
                const int NODES = 4096, HOPS = 20000;
                // every wave: NODES node addresses drawn uniformly over the pieces, linked in a random cycle
                std::vector<unsigned long long *> starts(W);
                std::vector<std::pair<unsigned long long, unsigned long long>> writes; // (address, value)
                writes.reserve((size_t)W * NODES);
                std::vector<unsigned long long> addr(NODES);
                for (int w = 0; w < W; w++) {
                    for (int n = 0; n < NODES; n++) {
                        const size_t pi = rng() % pieces.size();
                        const size_t off = (rng() % (psize[pi] / 64)) * 64 + (size_t)(w % 8) * 8; // (8 waves may share a sector: distinct words)
                        addr[n] = (unsigned long long)(pieces[pi] + off);
                    }
                    for (int n = 0; n < NODES; n++) writes.push_back({addr[n], addr[(n + 1) % NODES]});
                    starts[w] = (unsigned long long *)addr[0];
                }
                // the links go to the device as (address, value) pairs and are written there by a small kernel
                unsigned long long *d_a, *d_v;
                const size_t nw = writes.size();
                std::vector<unsigned long long> ha(nw), hv(nw);
                for (size_t i = 0; i < nw; i++) { ha[i] = writes[i].first; hv[i] = writes[i].second; }
                hipMalloc(&d_a, nw * 8); hipMalloc(&d_v, nw * 8);
                hipMemcpy(d_a, ha.data(), nw * 8, hipMemcpyHostToDevice); hipMemcpy(d_v, hv.data(), nw * 8, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(scat, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, d_a, d_v, nw);
                unsigned long long **d_s, *d_sink;
                hipMalloc(&d_s, W * 8); hipMalloc(&d_sink, W * 8);
                hipMemcpy(d_s, starts.data(), W * 8, hipMemcpyHostToDevice);
                hipDeviceSynchronize();
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(chase, dim3(W / 4), dim3(256), 0, 0, d_s, 2000, d_sink); // warm-up
                hipEventRecord(e0);
                hipLaunchKernelGGL(chase, dim3(W / 4), dim3(256), 0, 0, d_s, HOPS, d_sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                printf("%s  set %6.2f GB  %5d waves: %7.1f ns per dependent load (%zu allocations)\n", mode ? "pieces of 4-32 MB" : "one allocation   ", sg, W,
                       1e6 * ms / HOPS, pieces.size());
                fflush(stdout);
                hipFree(d_a); hipFree(d_v); hipFree(d_s); hipFree(d_sink);
            }
            for (auto q : pieces) hipFree(q);
        }
    }
    return 0;
}
