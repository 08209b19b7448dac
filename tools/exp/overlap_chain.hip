// Experiment (round 6): can a chain of DEPENDENT launches hide its per-launch fixed cost by alternating two streams and ordering the
// kernels with a device-side counter instead of the queue's barrier bit?  (hipExtAnyOrderLaunch is documented as unsupported on gfx9.)
// Layer k: out[i] = f(in[i-1], in[i], in[i+1]) over `n` floats, `blocks` blocks.  Mode 0: one stream.  Mode 1: streams A / B alternate, the
// consumer spins on the producer's done-counter after its prologue.  Each kernel uses <= 256 blocks at <= 2 resident per CU, and a kernel's
// predecessor-but-one has completed by stream order, so at most ONE resident kernel waits and it can never own all slots (no deadlock);
// the spin is bounded anyway.   hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_chain tools/exp/overlap_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void layer2_kernel(const float* __restrict__ in, float* __restrict__ out, long n, const float* __restrict__ w, int nw,
                                                     const unsigned* wait_ctr, unsigned wait_for, unsigned* done_ctr, int flagged, int* err) {
    __shared__ float ws[1024];
    for (int i = threadIdx.x; i < nw; i += 256) ws[i] = w[i];
    if (flagged && wait_for) {
        if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_for) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > 4000000) { *err = 1; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else __syncthreads();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += ws[(threadIdx.x + 37 * i) & (nw - 1)];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float a = in[i > 0 ? i - 1 : i], b = in[i], c = in[i + 1 < n ? i + 1 : i];
        out[i] = 0.25f * a + 0.5f * b + 0.25f * c + 1e-9f * s;
    }
    if (flagged) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int layers = 24, reps = 30;
    int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    unsigned* ctr; CK(hipMalloc(&ctr, 4 * 64)); 
    float* w; CK(hipMalloc(&w, 4096)); CK(hipMemset(w, 0, 4096));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const long sizes[] = {1L << 16, 1L << 20, 1L << 22, 1L << 24};       // 256 KB .. 64 MB per map
    const int blockss[] = {64, 128, 256};
    for (long n : sizes) for (int blocks : blockss) {
        float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
        std::vector<float> h(n); for (long i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) & 1023) / 1024.f;
        double res[2] = {0, 0}; float chk[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<float> ts;
            for (int r = 0; r < reps; ++r) {
                CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice));
                CK(hipMemsetAsync(ctr, 0, 4 * 64, sa)); CK(hipStreamSynchronize(sa));
                CK(hipEventRecord(e0, sa));
                if (mode == 1) { CK(hipEventRecord(ej, sa)); CK(hipStreamWaitEvent(sb, ej, 0)); }
                for (int k = 0; k < layers; ++k) {
                    hipStream_t s = (mode == 1 && (k & 1)) ? sb : sa;
                    const float* in = (k & 1) ? b : a; float* out = (k & 1) ? a : b;
                    hipLaunchKernelGGL(layer2_kernel, dim3(blocks), dim3(256), 0, s, in, out, n, w, 1024, ctr + (k > 0 ? k - 1 : 0), k > 0 ? (unsigned)blocks : 0u, ctr + k, mode, err);
                }
                if (mode == 1) { CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0)); }
                CK(hipEventRecord(e1, sa));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            res[mode] = ts[ts.size() / 2] * 1e3 / layers;
            CK(hipMemcpy(h.data(), a, 4 * 1, hipMemcpyDeviceToHost));
            std::vector<float> o(1024); CK(hipMemcpy(o.data(), a + n / 2, 4096, hipMemcpyDeviceToHost));
            for (float v : o) chk[mode] += v;
            for (long i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) & 1023) / 1024.f;
        }
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("n %9ld (%6.1f MB/map) blocks %3d: one stream %7.2f us/layer | two streams + flags %7.2f us/layer | checks %.6f %.6f %s err %d\n", n, n * 4 / 1e6, blocks,
               res[0], res[1], chk[0], chk[1], chk[0] == chk[1] ? "SAME" : "DIFFERENT", herr);
        fflush(stdout);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
