// Stand-alone: what a grid-wide barrier costs on an MI355X (8 XCDs, device-scope atomics through the fabric) -- the price
// of fusing the bundle adjustment's dependent kernels into one persistent launch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/microbench/grid_barrier.hip && /tmp/grid_barrier
// Every block: n rounds of { a little work; device-scope release fence; ticket; spin until all blocks of the round arrived }.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_barriers(unsigned *counter, int n_rounds, int fence, double *sink) {
    const unsigned nb = gridDim.x;
    double x = threadIdx.x;
    for (int r = 0; r < n_rounds; r++) {
        x = x * 1.0000001 + 1.0;                              // (something to order)
        if (fence) __threadfence();                           // release: this block's writes visible device-wide
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nb;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (x == -1.0) *sink = x;
}

__global__ void k_empty(double *sink) { if (threadIdx.x == 9999) *sink = 0; }

int main() {
    unsigned *counter; double *sink;
    hipMalloc(&counter, 4); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        for (int fence : {0, 1}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipMemset(counter, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                k_barriers<<<blocks, 256>>>(counter, 200, fence, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%4d blocks x 256 threads, %s: %.2f us per grid barrier\n", blocks, fence ? "with __threadfence" : "no fence", best * 1e3f / 200);
        }
    }
    // dependent empty launches on one stream, for comparison
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; i++) k_empty<<<1024, 256>>>(sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 dependent empty launches of 1024 blocks: %.2f us per launch\n", ms * 1e3f / 200);
    return 0;
}
