// ncc_arith_probe.hip -- which short sequences reproduce fl(sqrt(p)) and fl(a / q) bit for bit?
// Exhaustive over every positive normal float for the square root; 2^36 random pairs plus adversarial
// mantissas for the division.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-gpu-flush-denormals-to-zero tools/probes/ncc_arith_probe.hip -o /tmp/ncc_probe && /tmp/ncc_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "../../introtocomputervision_amd/csrc/ncc_arith.hpp"  // the two sequences stereo.hip uses (rows marked [used])

__device__ __forceinline__ float as_f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __float_as_uint(f); }

// LLVM's lowering without the denormal scaling: v_sqrt_f32 (1 ulp) then pick among s-1ulp, s, s+1ulp
__device__ __forceinline__ float sqrt_fix(float p) {
    const float s = __builtin_amdgcn_sqrtf(p);
    const float sm = as_f(as_u(s) - 1), sp = as_f(as_u(s) + 1);
    const float em = __builtin_fmaf(-sm, s, p), ep = __builtin_fmaf(-sp, s, p);
    float r = em <= 0.f ? sm : s;
    r = ep > 0.f ? sp : r;
    return r;
}
// rsq + two residual steps
__device__ __forceinline__ float sqrt_a2(float p) {
    const float r = __builtin_amdgcn_rsqf(p);
    float s = p * r;
    const float h = 0.5f * r;
    float d = __builtin_fmaf(-s, s, p);
    s = __builtin_fmaf(d, h, s);
    d = __builtin_fmaf(-s, s, p);
    return __builtin_fmaf(d, h, s);
}
// v_sqrt + one residual step with h = 0.5 * rsq
__device__ __forceinline__ float sqrt_b(float p) {
    float s = __builtin_amdgcn_sqrtf(p);
    const float h = 0.5f * __builtin_amdgcn_rsqf(p);
    const float d = __builtin_fmaf(-s, s, p);
    return __builtin_fmaf(d, h, s);
}
// Goldschmidt-coupled: s, h refined together, then a residual step
__device__ __forceinline__ float sqrt_g(float p) {
    const float r = __builtin_amdgcn_rsqf(p);
    float s = p * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-s, h, 0.5f);
    s = __builtin_fmaf(s, e, s);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-s, s, p);
    return __builtin_fmaf(d, h, s);
}

constexpr int NSQ = 5;
__global__ void sqrt_kernel(unsigned long long *bad, uint32_t *first_bad) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = 0x7f800000ull - 0x00800000ull;
    for (uint64_t i = n; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = 0x00800000u + (uint32_t)i;
        const float p = as_f(u);
        const float ref = sqrtf(p);
        const float v[NSQ] = {sqrt_fix(p), micv::ncc_sqrt(p), sqrt_a2(p), sqrt_b(p), sqrt_g(p)};
        const int ex = (int)(u >> 23) - 127;
        const int inr = ex >= -64 && ex <= 96;
        for (int k = 0; k < NSQ; k++)
            if (as_u(v[k]) != as_u(ref)) {
                atomicAdd(&bad[2 * k + inr], 1ull);
                if (inr) atomicMin(&first_bad[k], u);
            }
    }
}

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// one remainder step fewer
__device__ __forceinline__ float div_d0(float a, float q) {
    float r = __builtin_amdgcn_rcpf(q);
    const float e = __builtin_fmaf(-q, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float t = a * r;
    const float m = __builtin_fmaf(-q, t, a);
    return __builtin_fmaf(m, r, t);
}
// reciprocal from rsq(p), q = fl(sqrt(p)): no v_rcp
__device__ __forceinline__ float div_d2(float a, float q, float r0) {
    float r = r0;
    const float e = __builtin_fmaf(-q, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float t = a * r;
    float m = __builtin_fmaf(-q, t, a);
    t = __builtin_fmaf(m, r, t);
    m = __builtin_fmaf(-q, t, a);
    return __builtin_fmaf(m, r, t);
}

constexpr int NDV = 3;
// pairs (a, q = sqrt(p)): exponents of p in [-32, 78], |a| <= ~2 q (|ncc| <= 1 + slop; wider to be safe) down to 2^-80 q
__global__ void div_kernel(uint64_t seed, int iters, int mode, unsigned long long *bad) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        const uint64_t h = mix(seed + n * (uint64_t)iters + it), h2 = mix(h);
        uint32_t pm = (uint32_t)h & 0x7fffff, am = (uint32_t)(h >> 23) & 0x7fffff;
        if (mode == 1) {  // adversarial mantissas: all ones / one / few bits
            const uint32_t pat[8] = {0x7fffff, 0x7ffffe, 0, 1, 0x400000, 0x3fffff, 0x555555, 0x2aaaaa};
            pm = pat[h2 & 7] ^ ((h2 >> 3) & 1 ? (uint32_t)(h2 >> 40) & 3 : 0);
            if ((h2 >> 4) & 1) am = pat[(h2 >> 5) & 7] ^ ((uint32_t)(h2 >> 44) & 3);
        }
        const int pe = -32 + (int)((h >> 46) % 111);
        const float p = as_f(((uint32_t)(pe + 127) << 23) | pm);
        const float q = sqrtf(p);
        const int qe = (int)(as_u(q) >> 23) - 127;
        int ae = qe + 1 - (int)((h2 >> 8) % ((mode == 2) ? 3 : 82));
        if (ae < -120) ae = -120;
        const uint32_t sign = (uint32_t)(h2 >> 63) << 31;
        const float a = as_f(sign | ((uint32_t)(ae + 127) << 23) | am);
        const float ref = a / q;
        const float v[NDV] = {micv::ncc_div(a, q), div_d0(a, q), div_d2(a, q, __builtin_amdgcn_rsqf(p))};
        for (int k = 0; k < NDV; k++)
            if (as_u(v[k]) != as_u(ref)) atomicAdd(&bad[k], 1ull);
    }
}

int main() {
    unsigned long long *bad;
    uint32_t *first;
    (void)hipMalloc(&bad, 64 * sizeof(*bad));
    (void)hipMalloc(&first, 16 * sizeof(*first));
    (void)hipMemset(bad, 0, 64 * sizeof(*bad));
    (void)hipMemset(first, 0xff, 16 * sizeof(*first));
    sqrt_kernel<<<4096, 256>>>(bad, first);
    unsigned long long hb[64];
    uint32_t hf[16];
    (void)hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost);
    const char *sn[NSQ] = {"v_sqrt + 1-ulp pick (LLVM, unscaled)", "[used] rsq + 1 residual (ncc_sqrt)", "rsq + 2 residual", "v_sqrt + residual (h from rsq)",
                           "rsq + Goldschmidt + residual"};
    for (int k = 0; k < NSQ; k++)
        printf("sqrt %-42s mismatches: exponent in [-64, 96] %llu (first 0x%08x), outside %llu\n", sn[k], hb[2 * k + 1], hf[k], hb[2 * k]);
    const char *dn[NDV] = {"[used] rcp + 2 remainder steps (ncc_div)", "rcp + 1 remainder step", "rsq(p) as the reciprocal + 2 remainder steps"};
    for (int mode = 0; mode < 3; mode++) {
        (void)hipMemset(bad, 0, 64 * sizeof(*bad));
        const int iters = 1 << 12;
        for (int rep = 0; rep < (mode == 0 ? 16 : 4); rep++) div_kernel<<<4096, 256>>>(0x1234567ull * (rep + 1) + mode, iters, mode, bad);
        (void)hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
        const double n = (double)(mode == 0 ? 16 : 4) * 4096 * 256 * iters;
        for (int k = 0; k < NDV; k++) printf("div mode %d %-46s mismatches %llu of %.3g\n", mode, dn[k], hb[k], n);
    }
    return 0;
}
