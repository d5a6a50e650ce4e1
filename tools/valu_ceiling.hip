// tools/valu_ceiling.hip -- what one SIMD of gfx950 (MI355X) issues per cycle, measured: the ceiling bench.py prices
// `valu_issue_frac` against.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gpurun_out/valu_ceiling tools/valu_ceiling.hip && gpurun_out/valu_ceiling > profiles/r03_valu_ceiling.json
//
// For each instruction kind and each occupancy w = 1..8 waves per SIMD a kernel of CUs x w workgroups of 256 threads (one wave
// per SIMD each, no LDS) runs `iters` blocks of 64 instructions, either as 8 independent chains or as one dependent chain.
// Reported per (kind, chain form, w): wave-instructions per cycle and SIMD from the wall clock (hipEvents over the launch, the
// clock taken from the same launch's s_memtime span) and cycles per wave-instruction as one wave sees them.
// The correctly rounded binary32 division and square root the path tracer's parity contract fixes (-ffp-contract=off, IEEE
// `/` and sqrtf) are priced the same way in "plain fp32 instruction slots".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Kind { FMA32 = 0, MUL32, ADD32, MAX32, CNDMASK, FMA64, PKFMA32, RCP32, DIV32_IEEE, SQRT32_IEEE, CNDMASK_S, CMP_CND, CMP_S, AND32, MULLO, LSHLADD, MOV32, MIN32, MAX3, MIN3, MED3, BFI, PERM, SUB32, XOR32, ADDU32, LSHLREV, MADU24, CMP_E32, MAX32_NOIEEE, CVT_I32, MAX_I32, ADD3, CNDMASK_VCC_SET, MIX_MAX_FMA_1_1, MIX_MAX_FMA_1_3, MIX_MAX_FMA_3_1, MIX_CMPS_FMA_1_1, MIX_CNDS_FMA_1_1, MIX_CMPS_CNDS_FMA, MIX_SALU_FMA, N_KINDS };
static const char *kind_name[N_KINDS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32", "v_fma_f64", "v_pk_fma_f32", "v_rcp_f32", "ieee_div_f32 (a / b, C level)", "ieee_sqrt_f32 (sqrtf, C level)",
                                        "v_cndmask_b32_e64 (mask in an SGPR pair)", "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (pairs)", "v_cmp_lt_f32_e64 -> SGPR pair", "v_and_b32", "v_mul_lo_u32", "v_lshl_add_u32", "v_mov_b32",
                                        "v_min_f32", "v_max3_f32", "v_min3_f32", "v_med3_f32", "v_bfi_b32", "v_perm_b32", "v_sub_f32", "v_xor_b32", "v_add_u32", "v_lshlrev_b32", "v_mad_u32_u24", "v_cmp_lt_f32_e32 (vcc)", "v_max_f32, MODE.IEEE = 0", "v_cvt_i32_f32", "v_max_i32", "v_add3_u32", "v_cndmask_b32_e32, vcc set by s_mov before the loop",
                                        "mix v_max_f32 : v_fma_f32 = 1 : 1", "mix v_max_f32 : v_fma_f32 = 1 : 3", "mix v_max_f32 : v_fma_f32 = 3 : 1", "mix v_cmp_lt_f32_e64 : v_fma_f32 = 1 : 1", "mix v_cndmask_b32_e64 : v_fma_f32 = 1 : 1", "mix v_cmp_e64 -> its v_cndmask_e64 -> 2 v_fma_f32", "mix s_and_b64 : v_fma_f32 = 1 : 1 (64 + 64 per block, VALU counted)"};
static const int kind_ops_per_block[N_KINDS] = {64, 64, 64, 64, 64, 64, 64, 64, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64}; // the C-level kinds run 16 operations per block (each is a sequence of instructions)

template <int KIND, bool DEP>
__global__ __launch_bounds__(256) void k_issue(float *out, int iters, unsigned long long *span) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0f + (float)(threadIdx.x + k) * 1e-3f;
    const float b = 0.9999f + (float)threadIdx.x * 1e-9f, c = 1e-7f;
    double d[8];
    for (int k = 0; k < 8; ++k) d[k] = (double)a[k];
    const double db = (double)b, dc = (double)c;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p[8]; for (int k = 0; k < 8; ++k) { p[k].x = a[k]; p[k].y = a[k] + 0.5f; }
    const f2v pb = {b, b}, pc = {c, c};
    unsigned long long smask = 0x5555aaaa3333ccccull + (unsigned long long)blockIdx.x, sres = 0;
    asm volatile("" : "+s"(smask));
    if (KIND == MAX32_NOIEEE) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0"); // MODE.IEEE off for the rest of this wave
    if (KIND == CNDMASK_VCC_SET) asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int r = DEP ? 0 : (j & 7);
            if (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
            else if (KIND == MAX32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b) : );
            else if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[r]) : "v"(db), "v"(dc));
            else if (KIND == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(pb), "v"(pc));
            else if (KIND == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[r]));
            else if (KIND == CNDMASK_S) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "s"(smask));
            else if (KIND == CMP_CND) { if (j & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b) : ); else asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[r]), "v"(b) : "vcc"); }
            else if (KIND == CMP_S) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sres) : "v"(a[r]), "v"(b));
            else if (KIND == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(a[r]) : "v"(b));
            else if (KIND == MIN32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == SUB32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[r]) : "v"(c));
            else if (KIND == XOR32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[r]));
            else if (KIND == MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == CMP_E32) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[r]), "v"(b) : "vcc");
            else if (KIND == MAX32_NOIEEE) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == CVT_I32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[r]));
            else if (KIND == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[r]) : "v"(b));
            else if (KIND == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c));
            else if (KIND == CNDMASK_VCC_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[r]) : "v"(b) : );
            else if (KIND == MIX_MAX_FMA_1_1) { if (j & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == MIX_MAX_FMA_1_3) { if ((j & 3) == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == MIX_MAX_FMA_3_1) { if ((j & 3) != 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[r]) : "v"(b)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == MIX_CMPS_FMA_1_1) { if (j & 1) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sres) : "v"(a[r]), "v"(b)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == MIX_CNDS_FMA_1_1) { if (j & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "s"(smask)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == MIX_CMPS_CNDS_FMA) { if ((j & 3) == 0) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sres) : "v"(a[r]), "v"(b)); else if ((j & 3) == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "s"(sres)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == MIX_SALU_FMA) { asm volatile("s_and_b64 %0, %0, %1" : "+s"(sres) : "s"(smask) : "scc"); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r]) : "v"(b), "v"(c)); }
            else if (KIND == DIV32_IEEE) { if (j < 16) { a[DEP ? 0 : (j & 7)] = b / a[DEP ? 0 : (j & 7)]; asm volatile("" : "+v"(a[DEP ? 0 : (j & 7)])); } }
            else if (KIND == SQRT32_IEEE) { if (j < 16) { a[DEP ? 0 : (j & 7)] = __builtin_sqrtf(a[DEP ? 0 : (j & 7)]) + c; asm volatile("" : "+v"(a[DEP ? 0 : (j & 7)])); } }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) s += a[k] + (float)d[k] + p[k].x + p[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(sres & 1ull);
    if (threadIdx.x == 0) span[blockIdx.x] = t1 - t0;
}

typedef void (*KFn)(float *, int, unsigned long long *);
template <int K> KFn pick(bool dep) { return dep ? (KFn)k_issue<K, true> : (KFn)k_issue<K, false>; }
static KFn kernel_for(int kind, bool dep) {
    switch (kind) {
        case FMA32: return pick<FMA32>(dep); case MUL32: return pick<MUL32>(dep); case ADD32: return pick<ADD32>(dep); case MAX32: return pick<MAX32>(dep);
        case CNDMASK: return pick<CNDMASK>(dep); case FMA64: return pick<FMA64>(dep); case PKFMA32: return pick<PKFMA32>(dep); case RCP32: return pick<RCP32>(dep);
        case DIV32_IEEE: return pick<DIV32_IEEE>(dep); case SQRT32_IEEE: return pick<SQRT32_IEEE>(dep);
        case CNDMASK_S: return pick<CNDMASK_S>(dep); case CMP_CND: return pick<CMP_CND>(dep); case CMP_S: return pick<CMP_S>(dep); case AND32: return pick<AND32>(dep);
        case MULLO: return pick<MULLO>(dep); case LSHLADD: return pick<LSHLADD>(dep); case MOV32: return pick<MOV32>(dep);
        case MIN32: return pick<MIN32>(dep); case MAX3: return pick<MAX3>(dep); case MIN3: return pick<MIN3>(dep); case MED3: return pick<MED3>(dep); case BFI: return pick<BFI>(dep);
        case PERM: return pick<PERM>(dep); case SUB32: return pick<SUB32>(dep); case XOR32: return pick<XOR32>(dep); case ADDU32: return pick<ADDU32>(dep); case LSHLREV: return pick<LSHLREV>(dep);
        case MADU24: return pick<MADU24>(dep); case CMP_E32: return pick<CMP_E32>(dep); case MAX32_NOIEEE: return pick<MAX32_NOIEEE>(dep); case CVT_I32: return pick<CVT_I32>(dep);
        case MAX_I32: return pick<MAX_I32>(dep); case ADD3: return pick<ADD3>(dep); case CNDMASK_VCC_SET: return pick<CNDMASK_VCC_SET>(dep);
        case MIX_MAX_FMA_1_1: return pick<MIX_MAX_FMA_1_1>(dep); case MIX_MAX_FMA_1_3: return pick<MIX_MAX_FMA_1_3>(dep); case MIX_MAX_FMA_3_1: return pick<MIX_MAX_FMA_3_1>(dep);
        case MIX_CMPS_FMA_1_1: return pick<MIX_CMPS_FMA_1_1>(dep); case MIX_CNDS_FMA_1_1: return pick<MIX_CNDS_FMA_1_1>(dep); case MIX_CMPS_CNDS_FMA: return pick<MIX_CMPS_CNDS_FMA>(dep); default: return pick<MIX_SALU_FMA>(dep);
    }
}

int main(int argc, char **argv) {
    const int kind0 = argc > 1 ? std::atoi(argv[1]) : 0; // first kind to run
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *out; unsigned long long *span;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CHECK(hipMalloc(&span, (size_t)cus * 8 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"what\": \"wave64 instructions per cycle and SIMD, wall clock; cycles from s_memtime of the same launch\",\n \"rows\": [\n", prop.name, cus, cus * 4);
    bool first = true;
    for (int kind = kind0; kind < N_KINDS; ++kind)
        for (int dep = 0; dep < 2; ++dep)
            for (int w : {1, 2, 4, 8}) {
                const int iters = (kind == DIV32_IEEE || kind == SQRT32_IEEE) ? 2000 : 4000;
                KFn fn = kernel_for(kind, dep != 0);
                const int blocks = cus * w;
                hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 200, span); // warm-up (clocks, code)
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, iters, span);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                float ms = 0.0f; CHECK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<unsigned long long> sp(blocks);
                CHECK(hipMemcpy(sp.data(), span, (size_t)blocks * 8, hipMemcpyDeviceToHost));
                double mean = 0.0; unsigned long long mx = 0;
                for (auto v : sp) { mean += (double)v; mx = v > mx ? v : mx; }
                mean /= blocks;
                const double ops_per_wave = (double)iters * kind_ops_per_block[kind];
                const double wave_ops = ops_per_wave * blocks * 4.0;           // 4 waves per block
                const double cyc_per_op_wave = mean / ops_per_wave;              // as one wave sees it (s_memtime ticks = shader cycles)
                const double ghz = (double)mx / (ms * 1e6);                      // longest span / wall time: the clock of this launch
                const double ops_per_cyc_simd = wave_ops / ((double)mx * cus * 4.0);
                std::printf("%s  {\"inst\": \"%s\", \"chains\": \"%s\", \"waves_per_simd\": %d, \"per_cycle_per_simd\": %.4f, \"cycles_per_inst_one_wave\": %.2f, \"g_inst_per_s_chip\": %.1f, \"clock_ghz\": %.3f, \"ms\": %.3f}",
                            first ? "" : ",\n", kind_name[kind], dep ? "1 dependent" : "8 independent", w, ops_per_cyc_simd, cyc_per_op_wave, wave_ops / (ms * 1e-3) / 1e9, ghz, ms);
                first = false;
            }
    std::printf("\n ]}\n");
    return 0;
}
