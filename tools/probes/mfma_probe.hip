// mfma_probe.hip -- does v_mfma_f32_16x16x4_f32 / 32x32x2_f32 equal an ascending-k fmaf chain bit
// for bit, and how do the matrix pipe and the VALU share a SIMD?  (MFMA note.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// D(16x16) = C + A(16xK) B(Kx16), K = 4*steps; A row-major [16][K], B row-major [K][16].
__global__ void mm16(const float *A, const float *B, const float *C, float *D, int steps) {
    const int l = threadIdx.x, K = 4 * steps;
    v4f acc;
    for (int v = 0; v < 4; v++) acc[v] = C[(4 * (l / 16) + v) * 16 + (l % 16)];
    for (int s = 0; s < steps; s++) {
        const float a = A[(l % 16) * K + 4 * s + l / 16];
        const float b = B[(4 * s + l / 16) * 16 + (l % 16)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; v++) D[(4 * (l / 16) + v) * 16 + (l % 16)] = acc[v];
}

// D(32x32) = C + A(32xK) B(Kx32), K = 2*steps.
__global__ void mm32(const float *A, const float *B, const float *C, float *D, int steps) {
    const int l = threadIdx.x, K = 2 * steps;
    v16f acc;
    for (int v = 0; v < 16; v++) acc[v] = C[(8 * (v / 4) + 4 * (l / 32) + (v % 4)) * 32 + (l % 32)];
    for (int s = 0; s < steps; s++) {
        const float a = A[(l % 32) * K + 2 * s + l / 32];
        const float b = B[(2 * s + l / 32) * 32 + (l % 32)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; v++) D[(8 * (v / 4) + 4 * (l / 32) + (v % 4)) * 32 + (l % 32)] = acc[v];
}

// ---- pipe sharing: each wave runs NM MFMAs and NV independent v_fma per iteration ----------------
template <int NM, int NV, bool PK>
__global__ void mix(float *out, int iters, int mfma_waves_mask) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool do_m = (mfma_waves_mask >> w) & 1;
    v4f acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = 1.f + l * 1e-3f + i;
    const float a = 1.0001f + l * 1e-6f, b = 0.9999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int m = 0; m < NM; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int v = 0; v < NV; v++) x[v & 15] = fmaf(x[v & 15], a, b);
        }
        if (do_m && NV < 0) {
#pragma unroll
            for (int v = 0; v < -NV; v++) x[v & 15] = fmaf(x[v & 15], a, b);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) reinterpret_cast<unsigned long long *>(out + gridDim.x * blockDim.x)[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static float rnd() {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (float)((double)(rng >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

template <int M, int KSTEP>
static int check(const char *name, int steps, bool toeplitz) {
    const int K = KSTEP * steps;
    std::vector<float> A(M * K), B(K * M), C(M * M), D(M * M);
    for (auto &x : A) x = rnd() * 37.f;
    for (auto &x : B) x = rnd();
    for (auto &x : C) x = toeplitz ? 0.f : rnd();
    if (toeplitz) {  // B[k][n] = g[k - n] for 0 <= k-n <= 14, else 0: the row pass as a banded matrix
        float g[15];
        for (int i = 0; i < 15; i++) g[i] = expf(-(i - 7) * (i - 7) / 50.f) / 12.f;
        for (int k = 0; k < K; k++)
            for (int n = 0; n < M; n++) B[k * M + n] = (k - n >= 0 && k - n <= 14) ? g[k - n] : 0.f;
    }
    float *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
    if (M == 16) mm16<<<1, 64>>>(dA, dB, dC, dD, steps); else mm32<<<1, 64>>>(dA, dB, dC, dD, steps);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad_asc = 0, bad_desc = 0, bad_unfused = 0;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) {
            float asc = C[i * M + j], unf = C[i * M + j];
            for (int k = 0; k < K; k++) { asc = fmaf(A[i * K + k], B[k * M + j], asc); volatile float p = A[i * K + k] * B[k * M + j]; unf = unf + p; }
            float desc = C[i * M + j];
            for (int s = 0; s < steps; s++)
                for (int k = KSTEP - 1; k >= 0; k--) desc = fmaf(A[i * K + KSTEP * s + k], B[(KSTEP * s + k) * M + j], desc);
            const float d = D[i * M + j];
            bad_asc += memcmp(&d, &asc, 4) != 0;
            bad_desc += memcmp(&d, &desc, 4) != 0;
            bad_unfused += memcmp(&d, &unf, 4) != 0;
        }
    printf("%s steps=%d toeplitz=%d: mismatches vs ascending-fmaf %d, descending-within-step %d, unfused %d (of %d)\n", name, steps, (int)toeplitz, bad_asc, bad_desc, bad_unfused, M * M);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    return 0;
}

template <int NM, int NV, bool PK>
static void run_mix(const char *what, int mask, int threads) {
    const int blocks = 256 * 2, iters = 2000;
    float *out;
    hipMalloc(&out, (size_t)blocks * threads * 4 + blocks * (threads / 64) * 8 + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<NM, NV, PK><<<blocks, threads>>>(out, iters, mask);
    hipEventRecord(e0);
    mix<NM, NV, PK><<<blocks, threads>>>(out, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(blocks * (threads / 64));
    hipMemcpy(t.data(), out + (size_t)blocks * threads, t.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : t) mean += x; mean /= t.size();
    printf("%-64s %.3f ms, mean wave cycles/iter %.1f\n", what, ms, mean / iters);
    hipFree(out);
}

int main() {
    for (int steps : {1, 2, 8}) { check<16, 4>("16x16x4", steps, false); check<32, 2>("32x32x2", steps, false); }
    check<16, 4>("16x16x4", 8, true);
    check<32, 2>("32x32x2", 23, true);
    // 512 threads = 8 waves = 2 per SIMD, 2 workgroups per CU -> 4 waves per SIMD
    run_mix<8, 32, false>("all waves VALU: 32 v_fma / iter", 0x00, 512);
    run_mix<8, 32, false>("all waves MFMA: 8 mfma16x16x4 / iter", 0xFF, 512);
    run_mix<8, 32, false>("waves 0-3 MFMA(8), waves 4-7 VALU(32)", 0x0F, 512);
    run_mix<8, 32, false>("even waves MFMA(8), odd waves VALU(32)", 0x55, 512);
    run_mix<8, -32, false>("every wave: 8 mfma then 32 v_fma", 0xFF, 512);
    run_mix<8, -64, false>("every wave: 8 mfma then 64 v_fma", 0xFF, 512);
    run_mix<8, 64, false>("all waves VALU: 64 v_fma / iter", 0x00, 512);
    return 0;
}
