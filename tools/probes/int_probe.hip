// int_probe.hip -- issue cost of the integer / cross-lane VALU forms of the exact-sum stereo kernel
// (stereo_exact.hip): v_dot4_u32_u8 (SGPR and VGPR operands), v_mad_u32_u24, v_add/sub_u32, v_add3_u32,
// v_max_u32 (plain and DPP), v_permlane32/16_swap, v_cndmask, ds_read_b32.  WAVES waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/int_probe.hip -o /tmp/int_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned *out, int iters, unsigned sa) {
    __shared__ unsigned lds[1024];
    const unsigned l = threadIdx.x;
    lds[l & 1023] = l;
    __syncthreads();
    unsigned x[16];
    for (int i = 0; i < 16; i++) x[i] = l * 3 + i;
    unsigned a = l * 7 + 1, b = l ^ 0x55;
    const unsigned *lp = lds + (l & 63);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int v = 0; v < 32; v++) {
            unsigned &r = x[v & 15];
            if (KIND == 0) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r) : "s"(sa), "v"(a));
            if (KIND == 1) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r) : "v"(b), "v"(a));
            if (KIND == 2) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r) : "v"(b), "v"(a));
            if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
            if (KIND == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "s"(sa));
            if (KIND == 5) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(a));
            if (KIND == 6) asm volatile("v_max_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x[(v + 5) & 15]));
            if (KIND == 7) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r), "+v"(x[(v + 8) & 15]));
            if (KIND == 8) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r), "+v"(x[(v + 8) & 15]));
            if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(a));
            if (KIND == 10) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(size_t)lp), "n"(0));
            if (KIND == 11) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r) : "v"(a));
            if (KIND == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(a));
            if (KIND == 13) asm volatile("v_max_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x[(v + 5) & 15]));
            if (KIND == 14) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
        }
        if (KIND == 10) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned s = 0;
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int THREADS>
static void run(const char *what) {
    const int blocks = 256 * (1024 / THREADS) , iters = 4000;   // 1024 threads per CU = 4 waves per SIMD
    unsigned *out;
    hipMalloc(&out, (size_t)blocks * THREADS * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) k<KIND, THREADS><<<blocks, THREADS>>>(out, iters, 0x01020304u);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<KIND, THREADS><<<blocks, THREADS>>>(out, iters, 0x01020304u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double ns = ms * 1e6 / (4.0 * 32 * iters);
    printf("%-40s %.3f ms  %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", what, ms, ns, ns * 2.4);
    hipFree(out);
}

int main() {
    run<14, 512>("v_fma_f32");
    run<0, 512>("v_dot4_u32_u8 s,v,v");
    run<1, 512>("v_dot4_u32_u8 v,v,v");
    run<2, 512>("v_mad_u32_u24");
    run<3, 512>("v_add_u32");
    run<12, 512>("v_sub_u32");
    run<4, 512>("v_add3_u32 v,v,s");
    run<11, 512>("v_lshl_add_u32");
    run<5, 512>("v_max_u32");
    run<6, 512>("v_max_u32_dpp row_ror:8");
    run<13, 512>("v_max_u32_dpp quad_perm");
    run<7, 512>("v_permlane32_swap");
    run<8, 512>("v_permlane16_swap");
    run<9, 512>("v_cndmask_b32");
    run<10, 512>("ds_read_b32 (32 in flight)");
    return 0;
}
