// Microbenchmark (tools only, not part of the product): VALU issue cost per wave-instruction on this box, by opcode class
// and by waves per SIMD.  One workgroup of W waves per CU on every CU; each wave runs `iters` trips of 64 independent
// instructions (8 chains x 8); cycles from s_memtime of wave 0, SIMD throughput = cycles * (waves per SIMD) / instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/valu_ubench && /tmp/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2f q = {b0, b1};
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f m0 = {a0, a1, a2, a3}, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
    unsigned long long msk = 0x5555aaaa5555aaaaull;
    int addr = ((threadIdx.x * 7 + 3) & 63) * 4;
    unsigned sc = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {   // v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (OP == 1) {   // v_pk_fma_f32
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (OP == 2) {   // v_add_f32
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        } else if (OP == 3) {   // v_pk_add_f32
            asm volatile(REP8("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (OP == 4) {   // v_mov_b32
            asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 5) {   // v_pk_add_f32 with op_sel / neg modifiers (multiply-by-(-i) folded)
            asm volatile(REP8("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (OP == 6) {   // dependent chain of v_fma_f32 (latency)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                              "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (OP == 7) {   // v_cndmask_b32 (VOP2, vcc)
            asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "vcc");
        } else if (OP == 8) {   // v_cndmask_b32 e64 with an SGPR-pair mask
            asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(msk));
        } else if (OP == 9) {   // v_bfi_b32
            asm volatile(REP8("v_bfi_b32 %0, %9, %0, %8\n v_bfi_b32 %1, %9, %1, %8\n v_bfi_b32 %2, %9, %2, %8\n v_bfi_b32 %3, %9, %3, %8\n"
                              "v_bfi_b32 %4, %9, %4, %8\n v_bfi_b32 %5, %9, %5, %8\n v_bfi_b32 %6, %9, %6, %8\n v_bfi_b32 %7, %9, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
        } else if (OP == 10) {   // v_permlane32_swap_b32
            asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 11) {   // v_add_f32 with DPP row_ror:8
            asm volatile(REP8("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 12) {   // ds_bpermute_b32 (8 in flight, then wait)
            asm volatile(REP8("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                              "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));
        } else if (OP == 13) {   // v_log_f32
            asm volatile(REP8("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                              "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == 14) {   // v_mfma_f32_4x4x1_16b_f32, 8 independent chains
            asm volatile(REP8("v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n v_mfma_f32_4x4x1_16b_f32 %2, %8, %9, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %8, %9, %3\n"
                              "v_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n v_mfma_f32_4x4x1_16b_f32 %5, %8, %9, %5\n v_mfma_f32_4x4x1_16b_f32 %6, %8, %9, %6\n v_mfma_f32_4x4x1_16b_f32 %7, %8, %9, %7\n")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7) : "v"(b0), "v"(b1));
        } else if (OP == 15) {   // v_mfma_f32_4x4x1_16b_f32, 2 chains (dependent every other instruction)
            asm volatile(REP8("v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n")
                         : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7) : "v"(b0), "v"(b1));
        } else if (OP == 16) {   // s_nop 0
            asm volatile(REP8("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
        } else if (OP == 17) {   // alternating v_add_f32 / s_add_u32 (does scalar work ride along?)
            asm volatile(REP8("v_add_f32 %0, %0, %8\n s_add_u32 %9, %9, 1\n v_add_f32 %1, %1, %8\n s_add_u32 %9, %9, 1\n v_add_f32 %2, %2, %8\n s_add_u32 %9, %9, 1\n v_add_f32 %3, %3, %8\n s_add_u32 %9, %9, 1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(sc) : "scc");
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + m0.x + m1.y + m2.z + m3.w + m4.x + m5.y + m6.z + m7.w;
}

template <int OP>
void run(const char* name, int waves) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 8 * 16);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c[16] = {0};
    hipMemcpy(c, cyc, 8 * waves, hipMemcpyDeviceToHost);
    long long cmax = 0, cmin = 1LL << 62;
    for (int w = 0; w < waves; ++w) {
        cmax = c[w] > cmax ? c[w] : cmax;
        cmin = c[w] < cmin ? c[w] : cmin;
    }
    const double per = (double)cmax / (iters * 64.0), nw = (waves + 3) / 4;
    printf("%-26s %2d waves/CU: slowest wave %6.2f ticks/instr (fastest %5.2f) -> %5.2f ticks per SIMD issue; kernel %.1f us = %.2f ns per SIMD issue\n",
           name, waves, per, (double)cmin / (iters * 64.0), per / nw, ms * 1e3, ms * 1e6 / (iters * 64.0 * nw));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    // s_memtime runs at a fixed 100 MHz on some parts: calibrate against a known-latency loop is left to the reader; ratios matter
    for (int w : {4, 8, 16}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_add_f32", w);
        run<3>("v_pk_add_f32", w);
        run<5>("v_pk_add_f32 op_sel/neg", w);
        run<4>("v_mov_b32", w);
        run<7>("v_cndmask_b32", w);
        run<6>("v_fma_f32 dependent", w);
        run<8>("v_cndmask_b32_e64 sgpr", w);
        run<9>("v_bfi_b32", w);
        run<10>("v_permlane32_swap", w);
        run<11>("v_add_f32_dpp row_ror", w);
        run<12>("ds_bpermute x8 + wait", w);
        run<13>("v_log_f32", w);
        run<14>("mfma 4x4x1 8 chains", w);
        run<15>("mfma 4x4x1 2 chains", w);
        run<16>("s_nop 0", w);
        run<17>("v_add_f32 + s_add_u32 (x2)", w);
    }
    return 0;
}
