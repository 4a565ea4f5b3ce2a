// valu_issue_bench.hip -- what one vector instruction costs a gfx950 SIMD, per instruction class, EXEC mask and
// number of waves sharing the SIMD.  Measurement aid for the render kernels' roofline (they are bounded by vector
// issue, not by memory): the figures decide how SQ_ACTIVE_INST_VALU is turned into a utilisation.
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/valu_issue_bench scripts/valu_issue_bench.hip
//   ./scripts/valu_issue_bench                       one JSON line per (instruction, EXEC mask, waves per SIMD)
//   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES \
//       -- ./scripts/valu_issue_bench --pmc          fewer configurations, one kernel name per instruction class
//
// Every wave runs kIters iterations of 128 independent instructions of one class (eight accumulators, sixteen rounds)
// between two reads of the shader clock (s_memtime) and records where it ran.  Workgroups are 256 threads = one wave
// per SIMD of a CU, and the grid is 256 CUs x W workgroups, so W waves share every SIMD.  Cycles per wave-instruction
// as the SIMD sees them = (last end - first start among the waves of one SIMD) / (instructions those waves issued),
// median over the SIMDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define CHECK(call)                                                                             \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kIters = 1024;           // x 128 instructions = 131 072 wave-instructions per wave

enum Op {
    OP_ADD_F32, OP_MUL_F32, OP_FMA_F32, OP_MIN_F32, OP_MAX_F32, OP_MAX3_F32, OP_PK_ADD_F32, OP_PK_MUL_F32, OP_PK_FMA_F32,
    OP_CNDMASK_VCC, OP_CNDMASK_SGPR, OP_CMP_VCC, OP_CMP_SGPR, OP_MOV_B32, OP_ADD_U32, OP_LSHL_B32, OP_AND_B32, OP_MAD_U32_U24,
    OP_MUL_LO_U32, OP_CVT_F32_U32, OP_RCP_F32, OP_SQRT_F32, OP_DIV_SCALE, OP_DIV_FMAS, OP_DIV_FIXUP, OP_READFIRSTLANE,
    OP_SUB_F32, OP_FMAC_F32, OP_CNDMASK_VCC_E64, OP_CNDMASK_VCC_FRESH, OP_ADDC_VCC, OP_MOV_B64, OP_LSHL_ADD_U32, OP_MIN3_F32, OP_MED3_F32,
    OP_XOR_B32, OP_BFE_U32, OP_OR3_B32, OP_MIX_CND_ADD7, OP_MIX_CMP_CND, OP_MIX_CMP_CND64, OP_COUNT
};
static const char* kOpNames[OP_COUNT] = {
    "v_add_f32", "v_mul_f32", "v_fma_f32", "v_min_f32", "v_max_f32", "v_max3_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
    "v_cndmask_b32(vcc)", "v_cndmask_b32(sgpr)", "v_cmp_lt_f32(vcc)", "v_cmp_lt_f32(sgpr)", "v_mov_b32", "v_add_u32", "v_lshlrev_b32",
    "v_and_b32", "v_mad_u32_u24", "v_mul_lo_u32", "v_cvt_f32_u32", "v_rcp_f32", "v_sqrt_f32", "v_div_scale_f32", "v_div_fmas_f32",
    "v_div_fixup_f32", "v_readfirstlane_b32", "v_sub_f32", "v_fmac_f32", "v_cndmask_b32_e64(vcc)", "v_cndmask_b32(vcc rewritten by s_mov each 8)",
    "v_addc_co_u32(vcc)", "v_mov_b64", "v_lshl_add_u32", "v_min3_f32", "v_med3_f32", "v_xor_b32", "v_bfe_u32", "v_or3_b32",
    "mix: 1 v_cndmask_b32(vcc) + 7 v_add_f32", "mix: v_cmp_lt_f32(vcc) + v_cndmask_b32(vcc) pairs", "mix: v_cmp_lt_f32(sgpr) + v_cndmask_b32_e64(sgpr) pairs"};

// Eight instructions of one class on eight independent registers.
// (one asm statement per loop iteration: the compiler pads between asm statements with s_nop, which would be timed too)
#define E8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define E128(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt) E8(fmt)
#define BODY(fmt)                                                                                                         \
    asm volatile(E128(fmt)                                                                                                \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])         \
                 : "v"(k)                                                                                                 \
                 : "vcc", "s20", "s21", "s22", "s23")
#define BODY2(fmt)                                                                                                        \
    asm volatile(E128(fmt)                                                                                                \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])         \
                 : "v"(k2)                                                                                                \
                 : "vcc")

#define F_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define F_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define F_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define F_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define F_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define F_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define F_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_CNDV(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define F_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define F_CMPV(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define F_CMPS(i) "v_cmp_lt_f32 s[22:23], %" #i ", %8\n"
#define F_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define F_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define F_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define F_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %" #i "\n"
#define F_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define F_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define F_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define F_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define F_DSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define F_DFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_DFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_RFL(i) "v_readfirstlane_b32 s20, %" #i "\n"
#define F_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define F_FMAC(i) "v_fmac_f32 %" #i ", %8, %8\n"
#define F_CNDV64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define F_CNDVF(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define F_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define F_MOV64(i) "v_mov_b64 %" #i ", %8\n"
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define F_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define F_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define F_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %" #i "\n"
// mixed streams, eight instructions each
#define MIX_A "v_cndmask_b32 %0, %0, %8, vcc\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
#define MIX_B "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define MIX_C "v_cmp_lt_f32 s[22:23], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cmp_lt_f32 s[22:23], %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cmp_lt_f32 s[22:23], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cmp_lt_f32 s[22:23], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n"
#define M16(m) m m m m m m m m m m m m m m m m
#define BODYM(m)                                                                                                          \
    asm volatile(M16(m)                                                                                                   \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])         \
                 : "v"(k)                                                                                                 \
                 : "vcc", "s22", "s23")
// the VCC-reading select with VCC rewritten by the scalar unit before every eight of them
#define E8F(fmt) "s_mov_b64 vcc, s[20:21]\n" fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define E128F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt) E8F(fmt)
#define BODYF(fmt)                                                                                                        \
    asm volatile(E128F(fmt)                                                                                               \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])         \
                 : "v"(k)                                                                                                 \
                 : "vcc")

__device__ __forceinline__ unsigned long long shader_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
__device__ __forceinline__ unsigned long long real_clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

struct WaveRecord {
    unsigned long long t0, t1;             // shader clock before and after the loop
    unsigned long long ticks;              // the loop on the constant 100 MHz clock
    unsigned int hw_id, xcc_id;            // where the wave ran: HW_REG_HW_ID (SIMD, CU, SH, SE), HW_REG_XCC_ID
};

// One kernel name per instruction class (the counter passes are read per kernel name).
template <int OP>
__global__ void __launch_bounds__(256) issue_kernel(const unsigned long long mask, const float seed, WaveRecord* __restrict__ out, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + (float)(i + lane); p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float k = seed * 1.0000001f + 1.0f;
    const v2f k2 = v2f{k, k};
    asm volatile("s_mov_b64 vcc, exec\n s_mov_b64 s[20:21], exec" ::: "vcc", "s20", "s21");
    const bool on = (mask >> lane) & 1ull;
    __syncthreads();
    const unsigned long long t0 = shader_clock(), r0 = real_clock();
    if (on) {
        _Pragma("nounroll")
        for (int it = 0; it < kIters; it++) {
            if (OP == OP_ADD_F32) { BODY(F_ADD); }
            else if (OP == OP_MUL_F32) { BODY(F_MUL); }
            else if (OP == OP_FMA_F32) { BODY(F_FMA); }
            else if (OP == OP_MIN_F32) { BODY(F_MIN); }
            else if (OP == OP_MAX_F32) { BODY(F_MAX); }
            else if (OP == OP_MAX3_F32) { BODY(F_MAX3); }
            else if (OP == OP_PK_ADD_F32) { BODY2(F_PKADD); }
            else if (OP == OP_PK_MUL_F32) { BODY2(F_PKMUL); }
            else if (OP == OP_PK_FMA_F32) { BODY2(F_PKFMA); }
            else if (OP == OP_CNDMASK_VCC) { BODY(F_CNDV); }
            else if (OP == OP_CNDMASK_SGPR) { BODY(F_CNDS); }
            else if (OP == OP_CMP_VCC) { BODY(F_CMPV); }
            else if (OP == OP_CMP_SGPR) { BODY(F_CMPS); }
            else if (OP == OP_MOV_B32) { BODY(F_MOV); }
            else if (OP == OP_ADD_U32) { BODY(F_ADDU); }
            else if (OP == OP_LSHL_B32) { BODY(F_LSHL); }
            else if (OP == OP_AND_B32) { BODY(F_AND); }
            else if (OP == OP_MAD_U32_U24) { BODY(F_MAD24); }
            else if (OP == OP_MUL_LO_U32) { BODY(F_MULLO); }
            else if (OP == OP_CVT_F32_U32) { BODY(F_CVT); }
            else if (OP == OP_RCP_F32) { BODY(F_RCP); }
            else if (OP == OP_SQRT_F32) { BODY(F_SQRT); }
            else if (OP == OP_DIV_SCALE) { BODY(F_DSCALE); }
            else if (OP == OP_DIV_FMAS) { BODY(F_DFMAS); }
            else if (OP == OP_DIV_FIXUP) { BODY(F_DFIX); }
            else if (OP == OP_READFIRSTLANE) { BODY(F_RFL); }
            else if (OP == OP_SUB_F32) { BODY(F_SUB); }
            else if (OP == OP_FMAC_F32) { BODY(F_FMAC); }
            else if (OP == OP_CNDMASK_VCC_E64) { BODY(F_CNDV64); }
            else if (OP == OP_CNDMASK_VCC_FRESH) { BODYF(F_CNDVF); }
            else if (OP == OP_ADDC_VCC) { BODY(F_ADDC); }
            else if (OP == OP_MOV_B64) { BODY2(F_MOV64); }
            else if (OP == OP_LSHL_ADD_U32) { BODY(F_LSHLADD); }
            else if (OP == OP_MIN3_F32) { BODY(F_MIN3); }
            else if (OP == OP_MED3_F32) { BODY(F_MED3); }
            else if (OP == OP_XOR_B32) { BODY(F_XOR); }
            else if (OP == OP_BFE_U32) { BODY(F_BFE); }
            else if (OP == OP_OR3_B32) { BODY(F_OR3); }
            else if (OP == OP_MIX_CND_ADD7) { BODYM(MIX_A); }
            else if (OP == OP_MIX_CMP_CND) { BODYM(MIX_B); }
            else if (OP == OP_MIX_CMP_CND64) { BODYM(MIX_C); }
        }
    }
    const unsigned long long t1 = shader_clock(), r1 = real_clock();
    if (lane == 0) {
        WaveRecord r;
        r.t0 = t0; r.t1 = t1; r.ticks = r1 - r0;
        unsigned int hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r.hw_id = hw; r.xcc_id = xcc;
        out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
    }
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) sink[0] = s;                   // keeps the registers alive without a store in practice
}

struct SimdSpan { unsigned long long t0 = 0, t1 = 0; int waves = 0; };

struct MaskCase { const char* name; unsigned long long mask; int lanes; };
static const MaskCase kMasks[] = {
    {"all64", ~0ull, 64},
    {"low32", 0x00000000FFFFFFFFull, 32},
    {"high32", 0xFFFFFFFF00000000ull, 32},
    {"even32", 0x5555555555555555ull, 32},
    {"low16", 0x000000000000FFFFull, 16},
    {"16+16", 0x0000FFFF0000FFFFull, 32},
    {"low12", 0xFFFull, 12},
    {"low9", 0x1FFull, 9},
    {"low8", 0xFFull, 8},
    {"low4", 0xFull, 4},
    {"low2", 0x3ull, 2},
    {"lane0", 1ull, 1},
    {"lane63", 0x8000000000000000ull, 1},
    {"1of16x4", 0x0001000100010001ull, 4},
};

typedef void (*KernelFn)(const unsigned long long, const float, WaveRecord*, float*);
template <int OP> struct Table { static void fill(KernelFn* t) { t[OP] = issue_kernel<OP>; Table<OP + 1>::fill(t); } };
template <> struct Table<OP_COUNT> { static void fill(KernelFn*) {} };

int main(int argc, char** argv) {
    bool pmc = false;
    for (int i = 1; i < argc; i++) if (!strcmp(argv[i], "--pmc")) pmc = true;
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    KernelFn table[OP_COUNT];
    Table<0>::fill(table);
    const int max_blocks = cus * 8;
    WaveRecord* d_out = nullptr;
    float* d_sink = nullptr;
    CHECK(hipMalloc((void**)&d_out, sizeof(WaveRecord) * 4 * (size_t)max_blocks));
    CHECK(hipMalloc((void**)&d_sink, 64));
    std::vector<WaveRecord> h((size_t)max_blocks * 4);
    const int wave_counts_full[] = {8, 2, 1};
    const int wave_counts_pmc[] = {8};
    const int* wave_counts = pmc ? wave_counts_pmc : wave_counts_full;
    const int n_wave_counts = pmc ? 1 : 3;
    const int n_masks = (int)(sizeof(kMasks) / sizeof(kMasks[0]));
    // warm the clocks up
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(table[OP_FMA_F32], dim3(cus * 8), dim3(256), 0, 0, ~0ull, 1.0f, d_out, d_sink);
    CHECK(hipDeviceSynchronize());
    for (int op = 0; op < OP_COUNT; op++) {
        for (int mi = 0; mi < n_masks; mi++) {
            if (pmc && mi != 0 && mi != 1 && mi != 4 && mi != 11) continue;
            for (int wi = 0; wi < n_wave_counts; wi++) {
                const int W = wave_counts[wi];
                if (!pmc && W != 8 && mi > 1 && mi != 11) continue;          // lower occupancies for the full mask, half a wave and one lane only
                const int blocks = cus * W;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(table[op], dim3(blocks), dim3(256), 0, 0, kMasks[mi].mask, 1.0f, d_out, d_sink);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
                CHECK(hipMemcpy(h.data(), d_out, sizeof(WaveRecord) * 4 * (size_t)blocks, hipMemcpyDeviceToHost));
                // Waves of a SIMD issue oldest first: a younger wave does not even get through its prologue while two older
                // ones keep the SIMD busy, so one wave's own duration says little.  What counts is the span from the first
                // start to the last end among the waves that ran on one SIMD, over the instructions they issued together.
                std::map<unsigned long long, SimdSpan> simds;
                std::vector<double> ghz;
                for (int i = 0; i < blocks * 4; i++) {
                    const WaveRecord& r = h[(size_t)i];
                    const unsigned long long key = ((unsigned long long)r.xcc_id << 32) | (r.hw_id & 0xFF30u);   // SE, SH, CU, SIMD
                    SimdSpan& sp = simds[key];
                    if (sp.waves == 0 || r.t0 < sp.t0) sp.t0 = r.t0;
                    if (sp.waves == 0 || r.t1 > sp.t1) sp.t1 = r.t1;
                    sp.waves++;
                    if (r.ticks > 0) ghz.push_back((double)(r.t1 - r.t0) / (double)r.ticks * 0.1);
                }
                const double insts = (double)kIters * 128.0;
                std::vector<double> cyc;
                std::vector<int> per_simd;
                for (auto& kv : simds) {
                    cyc.push_back((double)(kv.second.t1 - kv.second.t0) / (insts * kv.second.waves));
                    per_simd.push_back(kv.second.waves);
                }
                std::sort(cyc.begin(), cyc.end());
                std::sort(ghz.begin(), ghz.end());
                std::sort(per_simd.begin(), per_simd.end());
                const double clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
                printf("{\"op\": \"%s\", \"exec\": \"%s\", \"lanes\": %d, \"waves_per_simd\": %d, \"wave_instructions_per_wave\": %.0f, "
                       "\"simds_seen\": %zu, \"waves_on_a_simd_min_max\": [%d, %d], \"cycles_per_wave_instruction_on_the_simd\": %.3f, "
                       "\"p5\": %.3f, \"p95\": %.3f, \"from_kernel_time\": %.3f, \"shader_clock_ghz\": %.3f, \"kernel_ms\": %.4f}\n",
                       kOpNames[op], kMasks[mi].name, kMasks[mi].lanes, W, insts, simds.size(), per_simd.front(), per_simd.back(),
                       cyc[cyc.size() / 2], cyc[cyc.size() / 20], cyc[cyc.size() - 1 - cyc.size() / 20],
                       ms * 1e-3 * clock * 1e9 / (insts * W), clock, ms);
                fflush(stdout);
            }
        }
    }
    CHECK(hipFree(d_out));
    CHECK(hipFree(d_sink));
    return 0;
}
