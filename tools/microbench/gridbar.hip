// Calibration: what a phase boundary costs inside ONE persistent kernel (grid barrier on a device-scope atomic) against
// a dependent launch, with real data exchanged across the boundary (every block reads what ANOTHER block wrote).
//   chain        : P dependent launches in a hipGraph (the baseline: the ~3.5-4 us floor)
//   fence        : plain loads / stores, agent-scope release fence + atomic arrive + spin + agent-scope acquire fence
//   sc1          : the exchanged data is stored / loaded with sc0 sc1 (write-through / L2-bypassing), no cache-wide fences
//   baronly      : barrier only, no data
// Every spin is bounded (SPIN_MAX polls, then the block sets an error flag and goes on), so the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int P = 200, SPIN_MAX = 1 << 15;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld_sc1(const f4* p) {
    f4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v;
}
__device__ __forceinline__ void st_sc1(f4* p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

__global__ void k_phase(const f4* a, f4* b, int nb) {
    const int src = (blockIdx.x * 37 + 11) % nb;
    f4 v = a[src * 256 + threadIdx.x];
    b[blockIdx.x * 256 + threadIdx.x] = v + 1.f;
}

// MODE 0 fence, 1 sc1, 2 barrier only
template <int MODE> __global__ void __launch_bounds__(256) k_persist(f4* a, f4* b, unsigned* ctr, int* err, int nb) {
    const int src = (blockIdx.x * 37 + 11) % nb;
    f4 *in = a, *out = b;
    for (int ph = 0; ph < P; ++ph) {
        if (MODE == 0) { f4 v = in[src * 256 + threadIdx.x]; if (v.x != (float)ph) *err = 2; out[blockIdx.x * 256 + threadIdx.x] = v + 1.f; }
        if (MODE == 1) { f4 v = ld_sc1(&in[src * 256 + threadIdx.x]); if (v.x != (float)ph) *err = 2; st_sc1(&out[blockIdx.x * 256 + threadIdx.x], v + 1.f); }
        // ---- grid barrier
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(ph + 1) * nb;
            int n = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++n > SPIN_MAX) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        f4* t = in; in = out; out = t;
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    f4 *A, *B; unsigned* ctr; int* err;
    CK(hipMalloc(&A, 256 * 256 * 16)); CK(hipMalloc(&B, 256 * 256 * 16)); CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(A, 0, 256 * 256 * 16)); CK(hipMemset(B, 0, 256 * 256 * 16)); CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nb : {32, 64, 128, 256}) {
        {   // chain
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < P; ++i) { if (i & 1) hipLaunchKernelGGL(k_phase, dim3(nb), dim3(256), 0, st, B, A, nb); else hipLaunchKernelGGL(k_phase, dim3(nb), dim3(256), 0, st, A, B, nb); }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int it = 0; it < 3; ++it) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < 10; ++it) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("blocks %3d  chain    %6.2f us per phase\n", nb, ms * 1e3 / (10 * P));
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(A, 0, 256 * 256 * 16, st));
                CK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL((k_persist<0>), dim3(nb), dim3(256), 0, st, A, B, ctr, err, nb);
                if (mode == 1) hipLaunchKernelGGL((k_persist<1>), dim3(nb), dim3(256), 0, st, A, B, ctr, err, nb);
                if (mode == 2) hipLaunchKernelGGL((k_persist<2>), dim3(nb), dim3(256), 0, st, A, B, ctr, err, nb);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best) best = ms;
            }
            int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            // the value every element must hold after P phases of +1 (checks the exchange was coherent)
            float chk[4]; CK(hipMemcpy(chk, (P & 1) ? B : A, 16, hipMemcpyDeviceToHost));
            printf("blocks %3d  %-8s %6.2f us per phase   err %d  v0 %.0f\n", nb, mode == 0 ? "fence" : mode == 1 ? "sc1" : "baronly", best * 1e3 / P, h, chk[0]);
            CK(hipMemset(err, 0, 4));
        }
    }
    return 0;
}
