// fp32 GEMM on the bf16 matrix cores by exact operand splitting ("bf16x3").
//
// The fp32-input MFMA runs at 1/16 of the bf16 MFMA rate (157 vs 2500 TF) and the path's GEMMs
// already sit at 83-88 % of that roof (gemm.hip).  Every fp32 number is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 significand bits, truncation split: a1 = top 16 bits of a,
// a2 = top 16 bits of a - a1, a3 = a - a1 - a2), so
//     a * b = sum_{i,j} a_i * b_j            (nine bf16 x bf16 products, each exact in fp32)
// The kernel evaluates the six products with weight >= 2^-16 (a1b1, a1b2, a2b1, a2b2, a1b3, a3b1)
// -- the three dropped ones are below 2^-24 |a b|, i.e. below half an fp32 ulp of the product -- on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the leading term and the five correction terms in
// separate accumulators that are added once at the end (so the correction terms do not perturb the
// rounding of the main chain).  6 x 32 = 192 matrix-pipe cycles per 32x32x16 block instead of 512.
//
// STATUS (round 1): experimental operator, exported and parity-tested (tests/test_gpu_ops.py:
// split3 exact; GEMM error within 4x of the fp32-MFMA kernel's against fp64) but NOT used by the
// engine.  Measured at C3: 162-165 TF fp32-equivalent for the GEMM alone (+0.16 ms to split a
// [163840,512] operand) against 131-138 TF for the fp32-MFMA kernel, i.e. 1.2x, not the 2.7x the MFMA
// count allows: with six operand planes a 128x128 tile needs 32 B/clk/CU of operand delivery at full
// matrix rate and the kernel saturates at the ~12 B/clk/CU this access pattern gets from L2.  The
// way forward is a 256x256 tile (16 B/clk) -- which with 6 planes only fits LDS at BK = 16 -- and
// producers that emit the planes; see DESIGN.md (h).
// Structure = the persistent streaming kernel of gemm.hip (same tile order, same epilogue): 4 waves,
// 128x128 tile, BK = 32, six bf16 planes per stage in LDS (rows padded to 80 B: conflict-free
// ds_read_b128), global loads two K-steps ahead in two register sets, every LDS / global
// instruction embedded between groups of four MFMAs, LDS-only barrier.
#include "gemm_common.h"

#include <type_traits>

namespace omnitok {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;                        // k elements per K-step
constexpr int SROW = 80;                       // LDS bytes per plane row (64 data + 16 pad)
constexpr int SPLANE = 128 * SROW;             // 10240 B per plane tile
constexpr int SSTAGE = 6 * SPLANE;             // 61440 B per stage (A planes 0-2, B planes 3-5)
constexpr int SPLIT_LDS_BYTES = 2 * SSTAGE;    // 122880

__host__ __device__ constexpr int prod_a(int p) { return (p == 2 || p == 3) ? 1 : (p == 5 ? 2 : 0); }
__host__ __device__ constexpr int prod_b(int p) { return (p == 1 || p == 3) ? 1 : (p == 4 ? 2 : 0); }

struct SplitGemmParams {
    GemmParams g;               // epilogue view (c, bias, residual, M, N, ld*, nbn, nbm, ntiles)
    const char *a3;             // A planes: plane i at a3 + i * a_plane (bytes), row-major [M, K] bf16
    const char *w3;             // W planes
    int64_t a_plane, w_plane;   // bytes
    int64_t lda_b, ldw_b;       // row pitch in bytes
};

// x[M, K] fp32 (row m at x + row(m) * ldx) -> three bf16 planes out[3][M][K]
__global__ __launch_bounds__(256) void split3_kernel(const float *__restrict__ x, int64_t ldx, int64_t M, int K,
                                                     int64_t rpg, int64_t gstride, int64_t goff,
                                                     unsigned short *__restrict__ out, int64_t plane_elems) {
    const int k4 = K >> 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * k4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / k4;
        const int c = (int)(idx % k4) * 4;
        int64_t row = m;
        if (rpg > 0) row = (m / rpg) * gstride + goff + (m % rpg);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + row * ldx + c);
        unsigned short p1[4], p2[4], p3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (note: __builtin_bit_cast applied directly to the vector element v[e] is miscompiled by
            // hipcc 7.2 -- it reads element 0 for every e -- so go through a scalar copy)
            const float xe = v[e];
            const unsigned u1 = __float_as_uint(xe) & 0xFFFF0000u;
            const float r1 = __fsub_rn(xe, __uint_as_float(u1));  // exact
            const unsigned u2 = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = __fsub_rn(r1, __uint_as_float(u2));  // exact, <= 8 significant bits
            p1[e] = (unsigned short)(u1 >> 16);
            p2[e] = (unsigned short)(u2 >> 16);
            p3[e] = (unsigned short)(__float_as_uint(r2) >> 16);
        }
        const int64_t o = m * K + c;
        *reinterpret_cast<uint2 *>(out + o) = make_uint2(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16));
        *reinterpret_cast<uint2 *>(out + plane_elems + o) = make_uint2(p2[0] | (p2[1] << 16), p2[2] | (p2[3] << 16));
        *reinterpret_cast<uint2 *>(out + 2 * plane_elems + o) =
            make_uint2(p3[0] | (p3[1] << 16), p3[2] | (p3[3] << 16));
    }
}

template <int FLAGS>
__global__ __launch_bounds__(256, 1) void gemm_bf16x3_persistent(SplitGemmParams sp) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const GemmParams &p = sp.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hi = lane >> 5;
    const int nk = p.K / SBK;
    if ((int)blockIdx.x >= p.ntiles) return;
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // loader: thread -> 16-byte chunk (tid & 3) of rows (tid >> 2) + 64 i, i < 2, of each plane
    const int lrow = tid >> 2, lch = tid & 3;
    const char *ap[2];
    const char *wp[2];
    auto set_ptrs = [&](int i) {
        const int lid = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, p.ntiles);
        int64_t tbm;
        int tbn;
        tile_coords(lid, p.nbm, p.nbn, p.gn, tbm, tbn);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int64_t gr = tbm * BM + lrow + 64 * r;
            if (gr > p.M - 1) gr = p.M - 1;
            ap[r] = sp.a3 + gr * sp.lda_b + lch * 16;
            int wr = tbn * BN + lrow + 64 * r;
            if (wr > p.N - 1) wr = p.N - 1;
            wp[r] = sp.w3 + (int64_t)wr * sp.ldw_b + lch * 16;
        }
    };
    const int st_off = lrow * SROW + lch * 16;  // + i * 64 * SROW + plane * SPLANE + stage * SSTAGE
    // two register sets of staged K-steps (set k&1 holds K-step k): piece = (plane 0..5) * 2 + i
    u32x4 gs[2][12];
    auto gpiece_load = [&](int set, int piece, int k0b) {
        const int pl = piece >> 1, i = piece & 1;
        const char *src = pl < 3 ? ap[i] + pl * sp.a_plane : wp[i] + (pl - 3) * sp.w_plane;
        gs[set][piece] = *reinterpret_cast<const u32x4 *>(src + k0b);
    };
    auto gpiece_store = [&](int set, int piece, int stage) {
        const int pl = piece >> 1, i = piece & 1;
        *reinterpret_cast<u32x4 *>(smem_c + stage * SSTAGE + pl * SPLANE + i * 64 * SROW + st_off) = gs[set][piece];
    };
    // fragments: [half][plane][row block] -> one MFMA operand (8 bf16) each
    u32x4 fa[2][3][2], fb[2][3][2];
    const int a_frag = (wm * 64 + r32) * SROW + hi * 32;
    const int b_frag = (wn * 64 + r32) * SROW + hi * 32;
    auto frag_load = [&](int half, int stage, int piece) {  // piece 0..11: (operand, plane, block)
        const int opnd = piece / 6, pl = (piece % 6) >> 1, blk = piece & 1;
        const char *base = smem_c + stage * SSTAGE + (opnd * 3 + pl) * SPLANE + blk * 32 * SROW + half * 16;
        if (opnd == 0)
            fa[half][pl][blk] = *reinterpret_cast<const u32x4 *>(base + a_frag);
        else
            fb[half][pl][blk] = *reinterpret_cast<const u32x4 *>(base + b_frag);
    };
    f32x16 acc_hi[2][2], acc_lo[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc_hi[i][j][r] = 0.0f;
                    acc_lo[i][j][r] = 0.0f;
                }
    };
    zero_acc();
    // the six kept products (A plane, B plane) = (0,0) (0,1) (1,0) (1,1) (0,2) (2,0);
    // product 0 goes to acc_hi, the rest to acc_lo
    auto mfma_group = [&](int half, int prod) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const u32x4 au = fa[half][prod_a(prod)][mb], bu = fb[half][prod_b(prod)][nb];
                const bf16x8 a = __builtin_bit_cast(bf16x8, au);
                const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
                if (prod == 0)
                    acc_hi[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc_hi[mb][nb], 0, 0, 0);
                else
                    acc_lo[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc_lo[mb][nb], 0, 0, 0);
            }
    };
    // One K-step (LDS stage `st`, K-step parity = register set PAR): half 0 consumes F[0] while it
    // prefetches F[1] and stores K-step g+1 (register set PAR^1) into the other stage; after the
    // barrier half 1 consumes F[1], prefetches F[0] of K-step g+1 and fetches K-step g+2 into set PAR.
    auto kstep = [&](auto par_c, int k0b) {
        constexpr int PAR = decltype(par_c)::value;
#pragma unroll
        for (int prod = 0; prod < 6; ++prod) {
            mfma_group(0, prod);
            frag_load(1, PAR, 2 * prod);
            frag_load(1, PAR, 2 * prod + 1);
            gpiece_store(PAR ^ 1, 2 * prod, PAR ^ 1);
            gpiece_store(PAR ^ 1, 2 * prod + 1, PAR ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
#pragma unroll
        for (int prod = 0; prod < 6; ++prod) {
            mfma_group(1, prod);
            frag_load(0, PAR ^ 1, 2 * prod);
            frag_load(0, PAR ^ 1, 2 * prod + 1);
            gpiece_load(PAR, 2 * prod, k0b);
            gpiece_load(PAR, 2 * prod + 1, k0b);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // prologue: K-step 0 -> LDS stage 0, K-step 1 -> register set 1 (nk is even and >= 2)
    set_ptrs(0);
#pragma unroll
    for (int pc = 0; pc < 12; ++pc) gpiece_load(0, pc, 0);
#pragma unroll
    for (int pc = 0; pc < 12; ++pc) gpiece_store(0, pc, 0);
#pragma unroll
    for (int pc = 0; pc < 12; ++pc) gpiece_load(1, pc, SBK * 2);
    lds_barrier();
#pragma unroll
    for (int pc = 0; pc < 12; ++pc) frag_load(0, 0, pc);

    for (int ti = 0; ti < my_tiles; ++ti) {
        for (int k = 0; k < nk - 2; k += 2) {
            kstep(P0{}, (k + 2) * SBK * 2);
            kstep(P1{}, (k + 3) * SBK * 2);
        }
        if (ti + 1 < my_tiles) set_ptrs(ti + 1);
        kstep(P0{}, 0);
        kstep(P1{}, SBK * 2);
        const int lid = xcd_remap((int)blockIdx.x + ti * (int)gridDim.x, p.ntiles);
        int64_t ebm;
        int ebn;
        tile_coords(lid, p.nbm, p.nbn, p.gn, ebm, ebn);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc_hi[i][j] += acc_lo[i][j];
        gemm_epilogue<FLAGS, false>(p, acc_hi, ebm, ebn, wm, wn, r32, hi);
        zero_acc();
    }
}

template <int FLAGS>
static int launch_split(SplitGemmParams sp, hipStream_t stream) {
    static bool attr = false;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        OT_HIP(hipGetDevice(&dev));
        OT_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    if (!attr) {
        OT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3_persistent<FLAGS>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_LDS_BYTES));
        attr = true;
    }
    int grid = n_cu < sp.g.ntiles ? n_cu : sp.g.ntiles;
    hipLaunchKernelGGL(gemm_bf16x3_persistent<FLAGS>, dim3(grid), dim3(256), SPLIT_LDS_BYTES, stream, sp);
    OT_LAUNCH_CHECK("gemm_bf16x3_persistent");
    return OMNITOK_OK;
}

}  // namespace omnitok

using namespace omnitok;

extern "C" int omnitok_split3(const float *x, int64_t ldx, int64_t M, int K, int64_t rows_per_group,
                              int64_t group_stride, int64_t group_offset, void *planes, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (M == 0) return OMNITOK_OK;
    OT_CHECK_ARG(x && planes && K % 4 == 0 && ldx % 4 == 0 && aligned16(x) && aligned16(planes),
                 "split3: bad arguments");
    const int64_t total = M * (K / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, M, K, rows_per_group,
                       group_stride, group_offset, static_cast<unsigned short *>(planes), M * (int64_t)K);
    OT_LAUNCH_CHECK("split3");
    return OMNITOK_OK;
}

extern "C" int omnitok_gemm_bf16x3(const void *a_planes, const void *w_planes, const float *bias,
                                   const float *residual, int64_t ldr, float *c, int64_t ldc, int64_t M, int N,
                                   int K, int flags, omnitok_stream_t stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    OT_CHECK_ARG(a_planes && w_planes && c, "gemm_bf16x3: null pointer");
    if (M == 0) return OMNITOK_OK;
    OT_CHECK_ARG(K % (2 * SBK) == 0, "gemm_bf16x3: K=%d must be a multiple of %d", K, 2 * SBK);
    OT_CHECK_ARG(N % 64 == 0, "gemm_bf16x3: N=%d must be a multiple of 64", N);
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_BIAS) || bias, "gemm_bf16x3: BIAS flag without bias");
    OT_CHECK_ARG(!(flags & OMNITOK_GEMM_RESIDUAL) || residual, "gemm_bf16x3: RESIDUAL flag without residual");
    SplitGemmParams sp;
    GemmParams &p = sp.g;
    p.a = nullptr; p.w = nullptr; p.bias = bias; p.residual = residual; p.c = c;
    p.lda = K; p.ldw = K; p.ldr = ldr; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.nbn = (N + BN - 1) / BN;
    p.nbm = (int)((M + BM - 1) / BM);
    const int64_t nt = (int64_t)p.nbm * p.nbn;
    OT_CHECK_ARG(nt < (1ll << 31), "gemm_bf16x3: grid too large");
    p.ntiles = (int)nt;
    p.a_rpg = 0; p.a_stride = 0; p.a_off = 0;
    p.gn = 8;
    p.trace = nullptr;
    sp.a3 = static_cast<const char *>(a_planes);
    sp.w3 = static_cast<const char *>(w_planes);
    sp.a_plane = M * (int64_t)K * 2;
    sp.w_plane = (int64_t)N * K * 2;
    sp.lda_b = (int64_t)K * 2;
    sp.ldw_b = (int64_t)K * 2;
    switch (flags) {
        case 0: return launch_split<0>(sp, stream);
        case OMNITOK_GEMM_BIAS: return launch_split<OMNITOK_GEMM_BIAS>(sp, stream);
        case OMNITOK_GEMM_RESIDUAL: return launch_split<OMNITOK_GEMM_RESIDUAL>(sp, stream);
        case OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL:
            return launch_split<OMNITOK_GEMM_BIAS | OMNITOK_GEMM_RESIDUAL>(sp, stream);
        case OMNITOK_GEMM_GEGLU:
            OT_CHECK_ARG(N % BN == 0, "gemm_bf16x3: GEGLU packed width %d must be a multiple of %d", N, BN);
            return launch_split<OMNITOK_GEMM_GEGLU>(sp, stream);
        default:
            set_error("gemm_bf16x3: unsupported epilogue flags %d", flags);
            return OMNITOK_ERR_INVALID;
    }
}
