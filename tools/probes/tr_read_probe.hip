// Probe (run once on an MI355X): which LDS element does ds_read_b64_tr_b16 return to lane l, slot j, as a function of the
// per-lane addresses?  And where does global_load_lds (16 B) put each lane's bytes?  Output: tools/probes/tr_read_probe.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((s16x4 __attribute__((address_space(3)))*)(p))

__global__ void k(const short* g, short* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    short* L = (short*)lds;
    for (int i = threadIdx.x; i < 8192; i += 64) L[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off;
    if (mode == 0) off = l * 4;                                   // each lane its own consecutive 8 bytes
    else if (mode == 1) off = (l & 15) * 64 + (l >> 4) * 4;       // lane -> row (stride 64 elements), 16-lane group -> column block
    else off = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16; // 4 rows x 16 cols block per 16-lane group, row stride 64
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(L + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
    if (mode == 3) {   // glds probe: global g[i] = i; lane copies its 16 bytes (8 shorts) g + l*8 (+ a per-lane permutation)
        __syncthreads();
        const short* src = g + ((l ^ 5) * 8);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(L + 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        for (int j = 0; j < 4; ++j) out[l * 4 + j] = L[1024 + l * 8 + j * 2];
    }
}
int main() {
    short *g, *o; short h[8192], r[256];
    hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(r));
    for (int i = 0; i < 8192; ++i) h[i] = (short)i;
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 32768, 0, g, o, mode);
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    }
    return 0;
}
