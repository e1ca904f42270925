// gfx950 check: the v_dot2c_f32_bf16 form of the bf16 hi / lo split against the convert / shift / subtract form, bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bits(bf2 v) { return __builtin_bit_cast(unsigned, v); }
template <int V>
__global__ void k(const float* in, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float v0 = in[2 * i], v1 = in[2 * i + 1];
    bf2 hi, lo;
    hi[0] = (__bf16)v0; hi[1] = (__bf16)v1;
    if constexpr (V == 0) {
        lo[0] = (__bf16)(v0 - (float)hi[0]); lo[1] = (__bf16)(v1 - (float)hi[1]);
    } else if constexpr (V == 1) {   // compiler-chosen operands (inline constants)
        bf2 m0, m1;
        m0[0] = (__bf16)-1.0f; m0[1] = (__bf16)0.0f; m1[0] = (__bf16)0.0f; m1[1] = (__bf16)-1.0f;
        lo[0] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(hi, m0, v0, false);
        lo[1] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(hi, m1, v1, false);
    } else {                         // multipliers forced into registers
        unsigned u0 = 0x0000bf80u, u1 = 0xbf800000u;
        asm volatile("" : "+v"(u0), "+v"(u1));
        const bf2 m0 = __builtin_bit_cast(bf2, u0), m1 = __builtin_bit_cast(bf2, u1);
        lo[0] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(hi, m0, v0, false);
        lo[1] = (__bf16)__builtin_amdgcn_fdot2_f32_bf16(hi, m1, v1, false);
    }
    out[2 * i] = bits(hi); out[2 * i + 1] = bits(lo);
}
__global__ void krelu(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(in[i]));
    out[i] = r;
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float m = (rand() / (float)RAND_MAX * 2 - 1);
        const int e = rand() % 40 - 30;
        h[i] = ldexpf(m, e);
    }
    h[0] = 0.0f; h[1] = -0.0f; h[2] = 1.0f; h[3] = -1.0f; h[4] = 1e-38f; h[5] = 3.0e38f;
    float *d, *dr; unsigned *o[3];
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&dr, n * 4);
    for (int v = 0; v < 3; ++v) (void)hipMalloc(&o[v], n * 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<0>, dim3(n / 2 / 256), dim3(256), 0, 0, d, o[0], n);
    hipLaunchKernelGGL(k<1>, dim3(n / 2 / 256), dim3(256), 0, 0, d, o[1], n);
    hipLaunchKernelGGL(k<2>, dim3(n / 2 / 256), dim3(256), 0, 0, d, o[2], n);
    hipLaunchKernelGGL(krelu, dim3(n / 256), dim3(256), 0, 0, d, dr, n);
    unsigned* r[3];
    for (int v = 0; v < 3; ++v) { r[v] = (unsigned*)malloc(n * 4); (void)hipMemcpy(r[v], o[v], n * 4, hipMemcpyDeviceToHost); }
    float* hr = (float*)malloc(n * 4);
    (void)hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost);
    for (int v = 1; v < 3; ++v) {
        int bad = 0, first = -1;
        for (int i = 0; i < n; ++i) if (r[v][i] != r[0][i]) { if (first < 0) first = i; ++bad; }
        printf("variant %d: %d of %d words differ from the reference split", v, bad, n);
        if (first >= 0) printf(" (first: word %d, in %g %g: ref %08x got %08x)", first, h[first & ~1], h[(first & ~1) + 1], r[0][first], r[v][first]);
        printf("\n");
    }
    int badr = 0;
    for (int i = 0; i < n; ++i) { const float e = h[i] > 0.0f ? h[i] : 0.0f; if (memcmp(&e, &hr[i], 4) != 0 && !(e == 0.0f && hr[i] == 0.0f)) ++badr; }
    printf("relu asm: %d of %d differ\n", badr, n);
    return 0;
}
