// Probe: what ds_read_b64_tr_b16 returns per lane (gfx950).  LDS holds lds[i] = i (16-bit); lane l reads 8 bytes at element offset off(l).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* offs) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + offs[threadIdx.x]));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
    short* out; int* offs;
    (void)hipHostMalloc((void**)&out, 64 * 4 * 2); (void)hipHostMalloc((void**)&offs, 64 * 4);
    // pattern A: lane l -> element 4*l (contiguous)
    // pattern B: row-major image [m][64 cols]: lane l in group g=l>>4: i=l&15 -> row (g*4 + i/4)? try both assignments
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, i = l & 15;
            if (pat == 0) offs[l] = 4 * l;
            if (pat == 1) offs[l] = (g * 4 + (i >> 2)) * 64 + (i & 3) * 4;   // lane i: row i/4, col quad i%4
            if (pat == 2) offs[l] = (g * 4 + (i & 3)) * 64 + (i >> 2) * 4;    // lane i: row i%4, col quad i/4
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, offs);
        (void)hipDeviceSynchronize();
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf(" lane %2d off %4d ->", l, offs[l]);
            for (int e = 0; e < 4; ++e) {
                int v = out[l * 4 + e];
                if (pat == 0) printf(" %4d", v); else printf(" (m%2d,c%2d)", v / 64, v % 64);
            }
            printf("\n");
        }
    }
    return 0;
}
