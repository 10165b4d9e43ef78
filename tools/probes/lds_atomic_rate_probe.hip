// LDS atomic / store rates on gfx950 (round 6, attention6.h's dQ tile): 8 waves per work-group, one work-group per CU, each lane issues
// N operations on its own address (row stride 65 dwords: conflict-free), timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_rate_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(512) void probe(uint64_t* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* f = reinterpret_cast<float*>(lds);
    unsigned* u = reinterpret_cast<unsigned*>(lds);
    unsigned long long* q = reinterpret_cast<unsigned long long*>(lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) f[i] = 0.f;
    __syncthreads();
    const int base = wave * 2048 + (lane & 31) * 65 + (lane >> 5) * 4;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int a = (base + (r & 3) + 8 * (r >> 2)) & 16383;
            if (MODE == 0) __hip_atomic_fetch_add(f + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(u + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add(q + (a >> 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) f[a] = (float)it;
            if (MODE == 4) { float v = f[a]; asm volatile("" :: "v"(v)); }
            if (MODE == 5) __hip_atomic_fetch_add(f + wave * 2048 + lane + 64 * (r & 15), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // lane-linear
        }
    }
    __syncthreads();
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (f[threadIdx.x] == 12345.f) out[0] = 0;
}

int main() {
    uint64_t* d;
    hipMalloc(&d, 256 * 8);
    const char* names[] = {"ds_add_f32 (rows x 65)", "ds_add_u32", "ds_add_u64", "ds_write_b32", "ds_read_b32", "ds_add_f32 lane-linear"};
    for (int mode = 0; mode < 6; ++mode) {
        const int iters = 64;
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 65536, 0, d, iters); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 65536, 0, d, iters); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 65536, 0, d, iters); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 65536, 0, d, iters); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 65536, 0, d, iters); break;
                default: hipLaunchKernelGGL(probe<5>, dim3(256), dim3(512), 65536, 0, d, iters); break;
            }
            hipDeviceSynchronize();
        }
        uint64_t h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double cyc = (double)h[7];
        // per CU: 8 waves x iters x 32 wave-instructions
        printf("%-28s %10.0f cycles for %d wave-instructions per CU = %7.1f cycles per wave-instruction (8 waves issuing)\n", names[mode], cyc, 8 * iters * 32,
               cyc / (8.0 * iters * 32));
    }
    return 0;
}
