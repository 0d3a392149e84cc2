// valu_rates.hip -- measured issue rates of the integer VALU instructions the orbfe kernels lean on (gfx950).
// Each kernel runs ITERS x 64 instructions of one kind on 8 independent register chains; grid = 256 CUs x 4 SIMDs x W.
// Prints wave-instructions per SIMD per cycle-equivalent: Gops/s (lane-ops) and cycles per wave-instruction per SIMD at
// the nominal 2.4 GHz.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2000

#define DEFKERNEL(name, ASM)                                                                      \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {                    \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,     \
                 a6 = a0 * 17, a7 = a0 * 19, b = seed * 2654435761u, c = seed + 12345;             \
        for (int i = 0; i < ITERS; i++) {                                                          \
            _Pragma("unroll") for (int j = 0; j < 8; j++) {                                        \
                asm volatile(ASM("%0") "\n" ASM("%1") "\n" ASM("%2") "\n" ASM("%3") "\n"           \
                             ASM("%4") "\n" ASM("%5") "\n" ASM("%6") "\n" ASM("%7")                \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                             : "v"(b), "v"(c));                                                    \
            }                                                                                      \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }

#define A_XOR(r) "v_xor_b32 " r ", " r ", %8"
#define A_BCNT(r) "v_bcnt_u32_b32 " r ", %8, " r
#define A_PKMAX(r) "v_pk_max_u16 " r ", " r ", %8"
#define A_MAX3(r) "v_max3_u32 " r ", " r ", %8, %9"
#define A_MAXU(r) "v_max_u32 " r ", " r ", %8"
#define A_ALIGNB(r) "v_alignbyte_b32 " r ", " r ", %8, 1"
#define A_PERM(r) "v_perm_b32 " r ", " r ", %8, %9"
#define A_DOT4(r) "v_dot4_u32_u8 " r ", " r ", %8, %9"
#define A_MED3(r) "v_med3_u32 " r ", " r ", %8, %9"
#define A_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9"
#define A_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8"
#define A_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8"
#define A_ADD(r) "v_add_u32 " r ", " r ", %8"
#define A_PKMUL(r) "v_pk_mul_lo_u16 " r ", " r ", %8"
#define A_FMA(r) "v_fma_f32 " r ", " r ", %8, %9"

#define A_PKMAXF(r) "v_pk_max_f16 " r ", " r ", %8"
#define A_PKMINF(r) "v_pk_min_f16 " r ", " r ", %8"
#define A_MAXF32(r) "v_max_f32 " r ", " r ", %8"
#define A_MAXF16(r) "v_max_f16 " r ", " r ", %8"
#define A_PKADDU(r) "v_pk_add_u16 " r ", " r ", %8"
#define A_PKSUBC(r) "v_pk_sub_u16 " r ", " r ", %8 clamp"
#define A_AND(r) "v_and_b32 " r ", " r ", %8"
#define A_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9"
#define A_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, 16"
#define A_DPP(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define A_PKADDF(r) "v_pk_add_f16 " r ", " r ", %8"
#define A_PKFMAF(r) "v_pk_fma_f16 " r ", " r ", %8, %9"
#define A_MAXI(r) "v_max_i32 " r ", " r ", %8"
#define A_MINU16(r) "v_min_u16 " r ", " r ", %8"
#define A_SAD(r) "v_sad_u8 " r ", " r ", %8, %9"
#define A_PKMAD(r) "v_pk_mad_u16 " r ", " r ", %8, %9"
#define A_DOT2(r) "v_dot2_u32_u16 " r ", " r ", %8, %9"
#define A_SUB(r) "v_sub_u32 " r ", " r ", %8"
#define A_LSHL(r) "v_lshlrev_b32 " r ", 3, " r
#define A_BFE(r) "v_bfe_u32 " r ", " r ", 8, 8"
#define A_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9"
#define A_OR3(r) "v_or3_b32 " r ", " r ", %8, %9"
#define A_CVTUB(r) "v_cvt_f32_ubyte0 " r ", " r
#define A_PKMAXI(r) "v_pk_max_i16 " r ", " r ", %8"
#define A_MAX3F(r) "v_max3_f32 " r ", " r ", %8, %9"
#define A_MAX3F16(r) "v_max3_f16 " r ", " r ", %8, %9"
#define A_PKMAX3F(r) "v_pk_maximum3_f16 " r ", " r ", %8, %9"
#define A_PKMIN3F(r) "v_pk_minimum3_f16 " r ", " r ", %8, %9"
#define A_MAX3U16(r) "v_max3_u16 " r ", " r ", %8, %9"
#define A_MAXIMUM3F32(r) "v_maximum3_f32 " r ", " r ", %8, %9"
#define A_PKMULF(r) "v_pk_mul_f32 " r ", " r ", %8"

DEFKERNEL(k_xor, A_XOR)
DEFKERNEL(k_pkmaxf, A_PKMAXF)
DEFKERNEL(k_pkminf, A_PKMINF)
DEFKERNEL(k_maxf32, A_MAXF32)
DEFKERNEL(k_maxf16, A_MAXF16)
DEFKERNEL(k_pkaddu, A_PKADDU)
DEFKERNEL(k_pksubc, A_PKSUBC)
DEFKERNEL(k_and, A_AND)
DEFKERNEL(k_andor, A_ANDOR)
DEFKERNEL(k_alignbit, A_ALIGNBIT)
DEFKERNEL(k_dpp, A_DPP)
DEFKERNEL(k_pkaddf, A_PKADDF)
DEFKERNEL(k_pkfmaf, A_PKFMAF)
DEFKERNEL(k_maxi, A_MAXI)
DEFKERNEL(k_minu16, A_MINU16)
DEFKERNEL(k_sad, A_SAD)
DEFKERNEL(k_pkmad, A_PKMAD)
DEFKERNEL(k_dot2, A_DOT2)
DEFKERNEL(k_sub, A_SUB)
DEFKERNEL(k_lshl, A_LSHL)
DEFKERNEL(k_bfe, A_BFE)
DEFKERNEL(k_add3, A_ADD3)
DEFKERNEL(k_or3, A_OR3)
DEFKERNEL(k_cvtub, A_CVTUB)
DEFKERNEL(k_pkmaxi, A_PKMAXI)
DEFKERNEL(k_max3f, A_MAX3F)
DEFKERNEL(k_max3f16, A_MAX3F16)
DEFKERNEL(k_pkmax3f, A_PKMAX3F)
DEFKERNEL(k_pkmin3f, A_PKMIN3F)
DEFKERNEL(k_max3u16, A_MAX3U16)
DEFKERNEL(k_maximum3f32, A_MAXIMUM3F32)
DEFKERNEL(k_bcnt, A_BCNT)
DEFKERNEL(k_pkmax, A_PKMAX)
DEFKERNEL(k_max3, A_MAX3)
DEFKERNEL(k_maxu, A_MAXU)
DEFKERNEL(k_alignb, A_ALIGNB)
DEFKERNEL(k_perm, A_PERM)
DEFKERNEL(k_dot4, A_DOT4)
DEFKERNEL(k_med3, A_MED3)
DEFKERNEL(k_mad24, A_MAD24)
DEFKERNEL(k_mullo, A_MULLO)
DEFKERNEL(k_lshlor, A_LSHLOR)
DEFKERNEL(k_add, A_ADD)
DEFKERNEL(k_pkmul, A_PKMUL)
DEFKERNEL(k_fma, A_FMA)

// ---- the Hamming scan's operand shapes (k_match.h: hamming256): the XOR takes the reference dword from an SGPR, the
// popcount accumulates in a VGPR chain.  k_xor_sgpr: v_xor_b32 v, s, v alone; k_hamming_chain: the scan's own sequence for one
// reference and eight independent queries-worth of chains -- 8 x (v_xor_b32 t, s, q ; v_bcnt_u32_b32 d, t, d) per chain
// register, counted as 16 instructions.
__global__ __launch_bounds__(256) void k_xor_sgpr(unsigned *out, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned sb = __builtin_amdgcn_readfirstlane(seed * 2654435761u);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            asm volatile("v_xor_b32 %0, %8, %0\nv_xor_b32 %1, %8, %1\nv_xor_b32 %2, %8, %2\nv_xor_b32 %3, %8, %3\n"
                         "v_xor_b32 %4, %8, %4\nv_xor_b32 %5, %8, %5\nv_xor_b32 %6, %8, %6\nv_xor_b32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "s"(sb));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

__global__ __launch_bounds__(256) void k_hamming_chain(unsigned *out, unsigned seed) {
    unsigned q0 = threadIdx.x + seed, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7;
    unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, t0, t1, t2, t3;
    const unsigned sb = __builtin_amdgcn_readfirstlane(seed * 2654435761u);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)   // 8 instructions per statement: 4 XOR + 4 BCNT, the scan's interleaving
            asm volatile("v_xor_b32 %4, %12, %8\nv_bcnt_u32_b32 %0, %4, %0\nv_xor_b32 %5, %12, %9\nv_bcnt_u32_b32 %1, %5, %1\n"
                         "v_xor_b32 %6, %12, %10\nv_bcnt_u32_b32 %2, %6, %2\nv_xor_b32 %7, %12, %11\nv_bcnt_u32_b32 %3, %7, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                         : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "s"(sb));
    }
    out[blockIdx.x * 256 + threadIdx.x] = d0 ^ d1 ^ d2 ^ d3;
}

// the same chain with the reference dword in a VGPR (as it would arrive from an LDS broadcast read): v_xor_b32 v, v, v is full rate
__global__ __launch_bounds__(256) void k_hamming_chain_vgpr(unsigned *out, unsigned seed) {
    unsigned q0 = threadIdx.x + seed, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7;
    unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, t0, t1, t2, t3;
    unsigned rv = seed * 2654435761u + (threadIdx.x >> 8);   // uniform value, but the compiler keeps it in a VGPR
    asm volatile("" : "+v"(rv));
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            asm volatile("v_xor_b32 %4, %12, %8\nv_bcnt_u32_b32 %0, %4, %0\nv_xor_b32 %5, %12, %9\nv_bcnt_u32_b32 %1, %5, %1\n"
                         "v_xor_b32 %6, %12, %10\nv_bcnt_u32_b32 %2, %6, %2\nv_xor_b32 %7, %12, %11\nv_bcnt_u32_b32 %3, %7, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                         : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(rv));
    }
    out[blockIdx.x * 256 + threadIdx.x] = d0 ^ d1 ^ d2 ^ d3;
}

typedef void (*kern_t)(unsigned *, unsigned);

int main(int argc, char **argv) {
    FILE *js = argc > 1 ? fopen(argv[1], "w") : nullptr;   // optional: the table as JSON (bench.py reads profiles/valu_rates.json)
    if (js) fprintf(js, "{\n \"clock_GHz\": 2.4,\n \"unit\": \"cycles per wave64 instruction per SIMD at the nominal clock\",\n \"rates\": {");
    bool first = true;
    unsigned *out;
    hipMalloc(&out, 256 * 4 * 16 * 256 * sizeof(unsigned) * 4);
    struct { const char *name; kern_t k; } ks[] = {
        {"v_xor_b32", k_xor}, {"v_bcnt_u32_b32", k_bcnt}, {"v_pk_max_u16", k_pkmax}, {"v_max3_u32", k_max3},
        {"v_max_u32", k_maxu}, {"v_alignbyte_b32", k_alignb}, {"v_perm_b32", k_perm}, {"v_dot4_u32_u8", k_dot4},
        {"v_med3_u32", k_med3}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_lshl_or_b32", k_lshlor},
        {"v_add_u32", k_add}, {"v_pk_mul_lo_u16", k_pkmul}, {"v_fma_f32", k_fma},
        {"v_pk_max_f16", k_pkmaxf}, {"v_pk_min_f16", k_pkminf}, {"v_max_f32", k_maxf32}, {"v_max_f16", k_maxf16},
        {"v_pk_add_u16", k_pkaddu}, {"v_pk_sub_u16 clamp", k_pksubc}, {"v_and_b32", k_and}, {"v_and_or_b32", k_andor},
        {"v_alignbit_b32", k_alignbit}, {"v_mov_b32_dpp", k_dpp}, {"v_pk_add_f16", k_pkaddf}, {"v_pk_fma_f16", k_pkfmaf},
        {"v_max_i32", k_maxi}, {"v_min_u16", k_minu16}, {"v_sad_u8", k_sad}, {"v_pk_mad_u16", k_pkmad},
        {"v_dot2_u32_u16", k_dot2}, {"v_sub_u32", k_sub}, {"v_lshlrev_b32", k_lshl}, {"v_bfe_u32", k_bfe},
        {"v_add3_u32", k_add3}, {"v_or3_b32", k_or3}, {"v_cvt_f32_ubyte0", k_cvtub}, {"v_pk_max_i16", k_pkmaxi},
        {"v_max3_f32", k_max3f}, {"v_max3_f16", k_max3f16}, {"v_pk_maximum3_f16", k_pkmax3f},
        {"v_pk_minimum3_f16", k_pkmin3f}, {"v_max3_u16", k_max3u16}, {"v_maximum3_f32", k_maximum3f32},
        {"v_xor_b32 v,s,v", k_xor_sgpr}, {"xor(s)+bcnt chain", k_hamming_chain},
        {"xor(v)+bcnt chain", k_hamming_chain_vgpr}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int wavesPerSimd[] = {4, 8};
    printf("%-18s %6s %12s %14s\n", "instruction", "w/SIMD", "Tlane-ops/s", "cyc/instr@2.4G");
    for (auto &k : ks)
        for (int w : wavesPerSimd) {
            const int blocks = 256 * w;   // 256 threads = 4 waves = one per SIMD of a CU
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 2u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instrPerWave = (double)ITERS * 64;
            const double laneOps = instrPerWave * 64.0 * blocks * 4;
            const double cyc = (ms * 1e-3 * 2.4e9) / (instrPerWave * w);
            printf("%-18s %6d %12.2f %14.2f\n", k.name, w, laneOps / (ms * 1e-3) / 1e12, cyc);
            if (js) {
                fprintf(js, "%s\n  \"%s @%d waves/SIMD\": %.3f", first ? "" : ",", k.name, w, cyc);
                first = false;
            }
        }
    if (js) {
        fprintf(js, "\n }\n}\n");
        fclose(js);
    }
    return 0;
}
