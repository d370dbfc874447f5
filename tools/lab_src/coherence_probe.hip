// Cross-XCD producer/consumer probe (lab tool): which publication protocol makes a payload visible before its flag?
//   hipcc --offload-arch=gfx950 -O3 -o coherence_probe coherence_probe.hip && ./coherence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int PAY = 1280;   // doubles per block and round (a tile: 5 fields x 256)
constexpr unsigned LIMIT = 1u << 20;

// MODE 0: sc1 (agent-scope atomic) payload stores + s_waitcnt 0 + barrier + sc1 flag; sc1 loads
// MODE 1: plain stores + __threadfence() by every thread + barrier + flag (release); consumer: acquire flag, __threadfence(), plain loads
// MODE 2: as 0, flag and payload at system scope
template <int MODE>
__global__ void __launch_bounds__(256) probe(double* pay, unsigned* flag, unsigned* consumed, unsigned* errors, unsigned* timeouts, int rounds)
{
    const unsigned G = gridDim.x, b = blockIdx.x, nb = (b + 1) % G;
    __shared__ unsigned s_to;
    if (threadIdx.x == 0) s_to = 0;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        // wait until the reader of my payload has consumed round r - 1
        if (threadIdx.x == 0 && r > 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&consumed[b * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > LIMIT) { s_to = 1; break; }
            }
        }
        __syncthreads();
        if (s_to) break;
        double* mine = pay + (size_t)b * PAY;
        for (int j = threadIdx.x; j < PAY; j += 256) {
            const double v = (double)r * 4096.0 + j;
            if (MODE == 1) mine[j] = v;
            else if (MODE == 2) __hip_atomic_store(mine + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else __hip_atomic_store(mine + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 1) __threadfence();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 1) __hip_atomic_store(&flag[b * 32], (unsigned)r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 2) __hip_atomic_store(&flag[b * 32], (unsigned)r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else __hip_atomic_store(&flag[b * 32], (unsigned)r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // consumer side: the neighbour's round r
            unsigned spins = 0;
            while (true) {
                unsigned f;
                if (MODE == 1) f = __hip_atomic_load(&flag[nb * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 2) f = __hip_atomic_load(&flag[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else f = __hip_atomic_load(&flag[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f >= (unsigned)r + 1) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > LIMIT) { s_to = 1; break; }
            }
        }
        __syncthreads();
        if (s_to) break;
        if (MODE == 1) __threadfence();
        const double* theirs = pay + (size_t)nb * PAY;
        unsigned bad = 0;
        for (int j = threadIdx.x; j < PAY; j += 256) {
            double v;
            if (MODE == 1) v = theirs[j];
            else if (MODE == 2) v = __hip_atomic_load(theirs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else v = __hip_atomic_load(theirs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += v != (double)r * 4096.0 + j;
        }
        if (bad) atomicAdd(errors, bad);
        __syncthreads();   // everybody has read
        if (threadIdx.x == 0) __hip_atomic_store(&consumed[nb * 32], (unsigned)r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0 && s_to) atomicAdd(timeouts, 1u);
}

int main(int argc, char** argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 1024, rounds = argc > 2 ? atoi(argv[2]) : 2000;
    double* pay; unsigned *flag, *consumed, *res;
    CHECK(hipMalloc(&pay, sizeof(double) * PAY * G));
    CHECK(hipMalloc(&flag, 128 * G)); CHECK(hipMalloc(&consumed, 128 * G)); CHECK(hipMalloc(&res, 8));
    for (int mode = 0; mode < 3; mode++) {
        CHECK(hipMemset(pay, 0, sizeof(double) * PAY * G)); CHECK(hipMemset(flag, 0, 128 * G)); CHECK(hipMemset(consumed, 0, 128 * G)); CHECK(hipMemset(res, 0, 8));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, pay, flag, consumed, res, res + 1, rounds);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, pay, flag, consumed, res, res + 1, rounds);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(G), dim3(256), 0, 0, pay, flag, consumed, res, res + 1, rounds);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[2]; CHECK(hipMemcpy(h, res, 8, hipMemcpyDeviceToHost));
        printf("mode %d: blocks %d rounds %d: stale doubles %u, timed-out blocks %u, %.3f ms (%.2f us/round)\n", mode, G, rounds, h[0], h[1], ms, ms * 1e3 / rounds);
    }
    return 0;
}
