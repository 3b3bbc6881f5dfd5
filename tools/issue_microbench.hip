// issue_microbench.hip -- what one gfx950 CU sustains per instruction class, measured.
//
// Settles which pipe bounds ring_search_kernel: the kernel issues
// ~2.0e9 VALU and ~1.35e9 SALU wave-instructions per launch, and the cost model of DESIGN.md
// depends on whether a wave64 integer VALU instruction holds its SIMD for 2 or 4 cycles when
// several waves are resident, and on how many scalar instructions a CU retires per cycle.
//
// Every kernel runs `iters` iterations of a 32-instruction body of one class (or a mix),
// written in inline asm so that nothing folds away, with W workgroups of 256 threads per CU
// (W = 1 -> one wave per SIMD, W = 8 -> eight).  Reported: cycles per instruction per SIMD
// (s_memtime ticks of the slowest wave / instructions of that SIMD's waves).
//
// build: hipcc -O2 --offload-arch=gfx950 tools/issue_microbench.hip -o /tmp/issue_microbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                               \
    do {                                                                                       \
        hipError_t r_ = (x);                                                                   \
        if (r_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_));                            \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

// 8 independent integer VALU ops on 8 accumulators (the cell test's v_sub / v_and_or mix)
#define VALU8                                                                                  \
    "v_sub_u32 %0, %0, %8\n v_and_or_b32 %1, %1, %8, %9\n v_sub_u32 %2, %2, %8\n"               \
    "v_and_or_b32 %3, %3, %8, %9\n v_sub_u32 %4, %4, %8\n v_and_or_b32 %5, %5, %8, %9\n"        \
    "v_sub_u32 %6, %6, %8\n v_and_or_b32 %7, %7, %8, %9\n"
#define SALU8                                                                                  \
    "s_add_u32 %0, %0, %8\n s_and_b32 %1, %1, %9\n s_add_u32 %2, %2, %8\n s_or_b32 %3, %3, %9\n" \
    "s_add_u32 %4, %4, %8\n s_and_b32 %5, %5, %9\n s_add_u32 %6, %6, %8\n s_or_b32 %7, %7, %9\n"
// 64-bit scalar mask algebra as the search kernel does it (s_and_b64 / s_ff1 / s_bcnt1)
#define SALU64_8                                                                               \
    "s_and_b64 %0, %0, %4\n s_ff1_i32_b64 %2, %0\n s_or_b64 %1, %1, %4\n s_bcnt1_i32_b64 %3, %1\n" \
    "s_andn2_b64 %0, %0, %4\n s_ff1_i32_b64 %2, %1\n s_xor_b64 %1, %1, %4\n s_bcnt1_i32_b64 %3, %0\n"

enum Kind { K_VALU, K_SALU, K_MIX, K_SALU64, K_CMP_READLANE, K_MIX_HALF, K_ADD64, K_ADD64_SGPR, K_SAD_U8, K_QSAD, K_MQSAD, K_PK_ADD_U16, K_VALU_Q1, K_VALU_H1, K_VALU_S16 };

template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(uint32_t iters, uint32_t *out,
                                                    unsigned long long *ticks)
{
    uint32_t v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5,
             v6 = v0 + 6, v7 = v0 + 7;
    const uint32_t va = 3u + (iters & 1u), vb = 0x10101u;
    uint32_t s0 = iters, s1 = iters + 1, s2 = iters + 2, s3 = iters + 3, s4 = iters + 4,
             s5 = iters + 5, s6 = iters + 6, s7 = iters + 7;
    const uint32_t sa = 7u, sb = 0xffff00ffu;
    unsigned long long m0 = 0x0f0f0f0f12345678ull + iters, m1 = ~m0;
    const unsigned long long mk = 0x5555aaaa3333ccccull;
    uint32_t r0 = 0, r1 = 0;
    unsigned long long w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3, w4 = w0 + 4, w5 = w0 + 5,
                       w6 = w0 + 6, w7 = w0 + 7;
    const unsigned long long wk = 0x0000010100000303ull + iters;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // partly filled waves: does a wave64 VALU instruction get cheaper when whole quarters (the
    // SIMD's 16-lane passes) or halves of EXEC are off?  Q1: lanes 0..15, H1: lanes 0..31,
    // S16: every fourth lane (16 lanes spread over all four passes)
    if constexpr (KIND == K_VALU_Q1 || KIND == K_VALU_H1 || KIND == K_VALU_S16) {
        const uint32_t ln = threadIdx.x & 63u;
        const bool act = KIND == K_VALU_Q1 ? ln < 16u : KIND == K_VALU_H1 ? ln < 32u : (ln & 3u) == 0u;
        if (act) {
            for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    asm volatile(VALU8
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5),
                                   "+v"(v6), "+v"(v7)
                                 : "v"(va), "v"(vb));
            }
        }
    } else
    for (uint32_t i = 0; i < iters; ++i) {
        if constexpr (KIND == K_VALU) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile(VALU8
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5),
                               "+v"(v6), "+v"(v7)
                             : "v"(va), "v"(vb));
        } else if constexpr (KIND == K_ADD64) { // 64-bit integer add (v_lshl_add_u64), all-VGPR
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n"
                             "v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                             "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n"
                             "v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                             : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                             : "v"(wk));
        } else if constexpr (KIND == K_ADD64_SGPR) { // the same with the first operand in an SGPR pair
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_lshl_add_u64 %0, %9, 0, %0\n v_lshl_add_u64 %1, %9, 0, %1\n"
                             "v_lshl_add_u64 %2, %9, 0, %2\n v_lshl_add_u64 %3, %9, 0, %3\n"
                             "v_lshl_add_u64 %4, %9, 0, %4\n v_lshl_add_u64 %5, %9, 0, %5\n"
                             "v_lshl_add_u64 %6, %9, 0, %6\n v_lshl_add_u64 %7, %9, 0, %7\n"
                             : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                             : "v"(wk), "s"(mk));
        } else if constexpr (KIND == K_SAD_U8) { // v_sad_u8: the windowed-SAD mode's inner instruction
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_sad_u8 %0, %0, %8, %0\n v_sad_u8 %1, %1, %8, %1\n v_sad_u8 %2, %2, %8, %2\n"
                             "v_sad_u8 %3, %3, %8, %3\n v_sad_u8 %4, %4, %8, %4\n v_sad_u8 %5, %5, %8, %5\n"
                             "v_sad_u8 %6, %6, %8, %6\n v_sad_u8 %7, %7, %8, %7\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                             : "v"(va));
        } else if constexpr (KIND == K_QSAD) { // packed quad SAD: 4 byte-shifted 4-byte SADs per instruction
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_qsad_pk_u16_u8 %0, %0, %8, %0\n v_qsad_pk_u16_u8 %1, %1, %8, %1\n"
                             "v_qsad_pk_u16_u8 %2, %2, %8, %2\n v_qsad_pk_u16_u8 %3, %3, %8, %3\n"
                             "v_qsad_pk_u16_u8 %4, %4, %8, %4\n v_qsad_pk_u16_u8 %5, %5, %8, %5\n"
                             "v_qsad_pk_u16_u8 %6, %6, %8, %6\n v_qsad_pk_u16_u8 %7, %7, %8, %7\n"
                             : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                             : "v"(va));
        } else if constexpr (KIND == K_MQSAD) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_mqsad_pk_u16_u8 %0, %0, %8, %0\n v_mqsad_pk_u16_u8 %1, %1, %8, %1\n"
                             "v_mqsad_pk_u16_u8 %2, %2, %8, %2\n v_mqsad_pk_u16_u8 %3, %3, %8, %3\n"
                             "v_mqsad_pk_u16_u8 %4, %4, %8, %4\n v_mqsad_pk_u16_u8 %5, %5, %8, %5\n"
                             "v_mqsad_pk_u16_u8 %6, %6, %8, %6\n v_mqsad_pk_u16_u8 %7, %7, %8, %7\n"
                             : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                             : "v"(va));
        } else if constexpr (KIND == K_PK_ADD_U16) { // two 16-bit adds per instruction
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n"
                             "v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n"
                             "v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                             : "v"(va));
        } else if constexpr (KIND == K_SALU) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile(SALU8
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5),
                               "+s"(s6), "+s"(s7)
                             : "s"(sa), "s"(sb)
                             : "scc");
        } else if constexpr (KIND == K_SALU64) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile(SALU64_8 : "+s"(m0), "+s"(m1), "+s"(r0), "+s"(r1) : "s"(mk) : "scc");
        } else if constexpr (KIND == K_MIX) { // 32 VALU + 24 SALU per iteration (0.75 per VALU)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile(VALU8
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5),
                               "+v"(v6), "+v"(v7)
                             : "v"(va), "v"(vb));
                if (k < 3)
                    asm volatile(SALU8
                                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5),
                                   "+s"(s6), "+s"(s7)
                                 : "s"(sa), "s"(sb)
                                 : "scc");
            }
        } else if constexpr (KIND == K_MIX_HALF) { // 32 VALU + 8 SALU
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile(VALU8
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5),
                               "+v"(v6), "+v"(v7)
                             : "v"(va), "v"(vb));
                if (k == 0)
                    asm volatile(SALU8
                                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5),
                                   "+s"(s6), "+s"(s7)
                                 : "s"(sa), "s"(sb)
                                 : "scc");
            }
        } else { // v_cmp -> SGPR pair, s_ff1, v_readlane: the ballot / broadcast idiom, x8
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned long long mm;
                uint32_t ln, bc;
                asm volatile("v_cmp_lt_u32 %0, %3, %4\n s_ff1_i32_b64 %1, %0\n s_and_b32 %1, %1, 63\n"
                             "s_nop 3\n v_readlane_b32 %2, %3, %1\n"
                             : "=s"(mm), "=s"(ln), "=s"(bc)
                             : "v"(v0), "v"(v1)
                             : "scc", "vcc");
                v1 += bc;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ s0 ^ s1 ^ s2 ^
                                          s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ (uint32_t)m0 ^ (uint32_t)m1 ^ r0 ^ r1 ^
                                          (uint32_t)((w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) >> 7);
    if ((threadIdx.x & 63u) == 0)
        ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
int run(const char *name, int wg_per_cu, double insts_per_iter, double salu_per_iter, int cus)
{
    const uint32_t iters = 20000;
    const int blocks = cus * wg_per_cu;
    uint32_t *out = nullptr;
    unsigned long long *ticks = nullptr;
    CHECK(hipMalloc((void **)&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc((void **)&ticks, (size_t)blocks * 4 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, 100u, out, ticks); // warm
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, iters, out, ticks);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    unsigned long long mx = 0;
    for (auto t : h) {
        sum += (double)t;
        mx = t > mx ? t : mx;
    }
    const double mean_ticks = sum / h.size();
    // s_memtime ticks at a fixed 100 MHz on this part: convert with the wall clock instead --
    // cycles = wall time x shader clock is what matters; report both
    const double waves_per_simd = wg_per_cu; // 4 waves per WG, one per SIMD
    const double inst_per_simd = insts_per_iter * iters * waves_per_simd;
    printf("%-14s waves/SIMD %d  wall %.3f ms  memtime mean %.0f max %llu ticks | per SIMD: %.3f ns/inst"
           " = %.2f cyc @2.4GHz (%.2f @ ticks)",
           name, wg_per_cu, ms, mean_ticks, mx, ms * 1e6 / inst_per_simd,
           ms * 1e6 / inst_per_simd * 2.4, (double)mx / inst_per_simd);
    if (salu_per_iter > 0)
        printf(" | SALU per CU: %.2f ns each", ms * 1e6 / (salu_per_iter * iters * waves_per_simd * 4));
    printf("\n");
    CHECK(hipFree(out));
    CHECK(hipFree(ticks));
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        if (run<K_VALU>("valu", w, 32, 0, cus)) return 1;
        if (run<K_SALU>("salu", w, 32, 32, cus)) return 1;
        if (run<K_SALU64>("salu64", w, 32, 32, cus)) return 1;
        if (run<K_MIX>("valu32+salu24", w, 56, 24, cus)) return 1;
        if (run<K_MIX_HALF>("valu32+salu8", w, 40, 8, cus)) return 1;
        if (run<K_CMP_READLANE>("cmp+ff1+rdlane", w, 40, 24, cus)) return 1;
        if (run<K_ADD64>("add64", w, 32, 0, cus)) return 1;
        if (run<K_ADD64_SGPR>("add64 sgpr", w, 32, 0, cus)) return 1;
        if (run<K_SAD_U8>("v_sad_u8", w, 32, 0, cus)) return 1;
        if (run<K_QSAD>("v_qsad_pk_u16", w, 32, 0, cus)) return 1;
        if (run<K_MQSAD>("v_mqsad_pk_u16", w, 32, 0, cus)) return 1;
        if (run<K_PK_ADD_U16>("v_pk_add_u16", w, 32, 0, cus)) return 1;
        if (run<K_VALU_Q1>("valu lanes0-15", w, 32, 0, cus)) return 1;
        if (run<K_VALU_H1>("valu lanes0-31", w, 32, 0, cus)) return 1;
        if (run<K_VALU_S16>("valu 16 spread", w, 32, 0, cus)) return 1;
    }
    return 0;
}
