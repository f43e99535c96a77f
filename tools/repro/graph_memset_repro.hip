// Minimal stand-alone check (no framework): does a hipMemsetAsync captured into a hipGraph run, with its byte value and its
// full size, on EVERY replay?  Sequence: capture {memset(buf, 0x7f, bytes); kernel counts words != 0x7f7f7f7f}; instantiate;
// replay (count must be 0); scribble over buf outside the graph; replay again (count must be 0 again); scribble, run
// other memsets (another size, value and stream) outside the graph; replay (0 again).
// build: hipcc --offload-arch=gfx950 -O2 -o graph_memset_repro graph_memset_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void count_bad(const unsigned *buf, size_t n, unsigned *bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (buf[i] != 0x7f7f7f7fu) atomicAdd(bad, 1u);
}
__global__ void scribble(unsigned *buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (unsigned)i;
}
int main(int argc, char **argv) {
    const size_t words = argc > 1 ? (size_t)atoll(argv[1]) : 655360;   // 65536 points x 10 slots, as the voxeliser's `top`
    unsigned *buf, *bad;
    CK(hipMalloc(&buf, words * 4));
    CK(hipMalloc(&bad, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipStream_t replay_stream = (argc > 2 && atoi(argv[2])) ? (hipStream_t)0 : s;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(bad, 0, 4, s));
    CK(hipMemsetAsync(buf, 0x7f, words * 4, s));
    hipLaunchKernelGGL(count_bad, dim3(256), dim3(256), 0, s, buf, words, bad);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int rc = 0;
    unsigned *other;
    CK(hipMalloc(&other, words * 4));
    for (int round = 0; round < 3; ++round) {
        if (round) {   // something else writes the buffer between two replays ...
            hipLaunchKernelGGL(scribble, dim3(256), dim3(256), 0, s, buf, words);
            CK(hipStreamSynchronize(s));
            if (round == 2) {   // ... and a memset of ANOTHER size and value runs outside the graph (default stream)
                CK(hipMemsetAsync(other, 0x11, words * 2 - 1234, 0));
                CK(hipMemsetAsync(buf, 0x22, words * 2, 0));
                CK(hipMemsetAsync(other, 0, words * 4, 0));   // (what tensor.zero_() issues)
                CK(hipDeviceSynchronize());
            }
        }
        CK(hipGraphLaunch(ge, replay_stream));   // (PyTorch replays on its current stream: the null stream by default)
        CK(hipDeviceSynchronize());
        unsigned h = 0;
        CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("replay %d%s: %u of %zu words are not the memset pattern\n", round, round == 2 ? " (after a scribble and eager memsets)" : round ? " (after a scribble)" : "", h, words);
        if (h) rc = 1;
    }
    printf(rc ? "FAIL: the captured memset did not restore the buffer\n" : "ok\n");
    return rc;
}
