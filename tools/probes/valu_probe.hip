// valu_probe.hip -- issue cost of the f32 VALU forms the LK kernel uses, 4 waves per SIMD
// (512-thread workgroups, 2 per CU): v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_mov_b32, f64 fma.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512, 4) void k(float *out, int iters) {
    const int l = threadIdx.x;
    v2f x[16];
    for (int i = 0; i < 16; i++) x[i] = (v2f){1.f + l * 1e-3f + i, 2.f + i};
    v2f a = {1.0001f + l * 1e-6f, 0.9997f}, b = {0.9999f, 1.0002f};
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = 1.0 + i + l * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 32; v++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 15].x) : "v"(a.x), "v"(b.x));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 15]) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[v & 15]) : "v"(a));
            if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "+v"(x[v & 15].x) : "v"(a.x));
            if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[v & 7]) : "v"(d[(v + 1) & 7]), "v"(d[(v + 2) & 7]));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(x[v & 15]) : "v"(a), "v"(b));
            if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[v & 15]) : "v"(a));
            if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 15].x) : "v"(a.x));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += x[i].x + x[i].y;
    for (int i = 0; i < 8; i++) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *what) {
    const int blocks = 512, iters = 4000;
    float *out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) k<KIND><<<blocks, 512>>>(out, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<KIND><<<blocks, 512>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // 4 waves per SIMD x 32 instructions x iters per kernel
    const double ns = ms * 1e6 / (4.0 * 32 * iters);
    printf("%-44s %.3f ms  %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", what, ms, ns, ns * 2.4);
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<5>("v_pk_fma_f32 with op_sel (skewed form)");
    run<2>("v_pk_mul_f32");
    run<6>("v_pk_add_f32");
    run<7>("v_add_f32");
    run<3>("v_mov_b32");
    run<4>("v_fma_f64");
    return 0;
}
