// How fast is ONE device-scope counter when thousands of wavefronts on all XCDs bump it?  (Pricing a dense completion queue: one atomicAdd per
// published game, 4096 per wave of ~130 us.)  Each wavefront: `iters` x { atomicAdd(counter, 1) by lane 0 (agent scope, returning), then a
// system-scope store to queue[pos] }, with `gap` s_sleep units between them.  Prints the mean latency a wave sees and the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_probe.hip -o tools/probes/atomic_probe && tools/probes/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned* ctr, unsigned long long* queue, unsigned long long* lat, int iters, int gap) {
    unsigned long long t = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned pos = 0;
        const unsigned long long t0 = wall_clock64();
        if (threadIdx.x == 0) pos = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos = __builtin_amdgcn_readfirstlane(pos);
        t += wall_clock64() - t0;
        if (threadIdx.x == 0) __hip_atomic_store(queue + (pos & 0xFFFFF), (unsigned long long)pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) lat[blockIdx.x] = t;
}
int main() {
    unsigned* ctr; unsigned long long *queue, *lat;
    hipMalloc(&ctr, 4); hipMalloc(&queue, 8 << 20); hipMalloc(&lat, 8 * 8192);
    for (int waves : {256, 1024, 4096}) for (int gap : {0, 16, 64}) {
        const int iters = 64;
        hipMemset(ctr, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_probe, dim3(waves), dim3(64), 0, 0, ctr, queue, lat, 4, 0);       // warm
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe, dim3(waves), dim3(64), 0, 0, ctr, queue, lat, iters, gap);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(waves); hipMemcpy(h.data(), lat, 8 * waves, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("waves %5d gap %3d: kernel %.1f us, %.1f atomics/us chip-wide, mean latency per atomic %.2f us\n", waves, gap, ms * 1e3, (double)waves * iters / (ms * 1e3),
               s / waves / iters / 100.0);
    }
    return 0;
}
