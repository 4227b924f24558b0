// gemm_split.hip — fp32-accurate GEMMs on the bf16 matrix pipe (gfx950): every fp32 operand is split into three bf16 planes
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)          (3 x 8 mantissa bits = fp32's 24)
// and a product a*b is formed from the six plane products that matter,
//     a*b = hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi)  +  O(2^-24 |a b|)            ("bf16x6"),
// each a `v_mfma_f32_32x32x16_bf16` with fp32 accumulation.  bf16 x bf16 products are exact in fp32, so the only roundings are those of
// the fp32 accumulator - as in the fp32-input MFMA path - and the dropped plane products (mid*lo, lo*mid, lo*lo: <= 2^-23 relative per
// product, random sign).  Six bf16 MFMAs cost 6 x 32 cycles per 32x32x16 block against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same
// K = 16: 2.67x the fp32-MFMA rate (419 instead of 157 TFLOP/s of fp32-equivalent multiply-adds).
//
// Used by the Winograd F(2x2,3x3) path of the deep 3x3 layers (model/yolo2.py:76-113; algo = Y2_ALGO_WINOGRAD_SPLIT): the 16 GEMMs
// M[p] = V[p] (T x Cin) * U[p]^T (Cin x Cout) of wino.hip.  The transforms stay fp32; the input transform writes V directly as planes
// (wino_input_split_kernel), the filter transform U is split once per weight version (y2_split_bf16x3).
//
// Kernel: 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each = 2 x 2 blocks of 32 x 32), K-slab 32 (64 B per row and plane), 3-deep LDS ring
// of 48 KB stages (3 planes x (128 + 128) rows x 64 B) filled with LDS-DMA (`buffer_load_dwordx4 ... lds`), one s_barrier per slab,
// DMA of slab s+2 issued between the MFMAs of slab s.  LDS rows are unpadded 64-B rows; the 16-B chunk a lane fetches is XOR-swizzled
// with (row >> 2) & 3 and the fragment reads apply the same involution: every ds_read_b128 lane group hits 16 distinct 16-B slots.
// Per slab and wave: 24 ds_read_b128, 48 MFMAs (1536 cycles), 12 DMA instructions.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// round-to-nearest-even fp32 -> bf16 bits (finite inputs; Inf stays Inf)
__device__ __forceinline__ unsigned bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_value(unsigned b) { return __uint_as_float(b << 16); }

__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = bf16_bits(x);
    const float r1 = x - bf16_value(hi);          // exact: the difference of two floats within a factor 2^-8
    mid = bf16_bits(r1);
    const float r2 = r1 - bf16_value(mid);        // exact
    lo = bf16_bits(r2);
}

// Two fp16 planes of x * scale: hi = fp16(x), lo = fp16(x - hi): 2 x 11 mantissa bits + sign of the residual = fp32's 24 as long as lo is a normal
// fp16 number; for small x the residual's subnormal quantum (2^-24 absolute) bounds the error instead - which is why operands are scaled into
// fp16's range by fixed powers of two (exact; undone in the GEMM epilogue).  Overflow: |x * scale| > 65504 would become Inf - see y2_split_f16x2.
__device__ int y2_f16_overflow_flag = 0;      // set (never cleared by kernels) when a scaled operand left fp16's range: y2_split_f16_overflow

__device__ __forceinline__ void split2h(float x, unsigned& hi, unsigned& lo) {
    if (!(fabsf(x) <= 65504.f)) y2_f16_overflow_flag = 1;      // (also catches NaN; plain store, every writer writes the same value)
    const _Float16 h = (_Float16)x;                 // v_cvt_f16_f32: round to nearest even
    const float r = x - (float)h;                    // exact
    const _Float16 l = (_Float16)r;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

__global__ __launch_bounds__(256) void split2h_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n4, long long plane, float scale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        u16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned a, b;
            split2h(v[e] * scale, a, b);
            h[e] = (unsigned short)a; l[e] = (unsigned short)b;
        }
        reinterpret_cast<u16x4*>(dst)[i] = h;
        reinterpret_cast<u16x4*>(dst + plane)[i] = l;
    }
}

// dst[p][i] = plane p of src[i], p = 0 (hi), 1 (mid), 2 (lo); n % 4 == 0
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n4, long long plane) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        u16x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned a, b, c;
            split3(v[e], a, b, c);
            h[e] = (unsigned short)a; m[e] = (unsigned short)b; l[e] = (unsigned short)c;
        }
        reinterpret_cast<u16x4*>(dst)[i] = h;
        reinterpret_cast<u16x4*>(dst + plane)[i] = m;
        reinterpret_cast<u16x4*>(dst + 2 * plane)[i] = l;
    }
}

// Winograd input transform V = B^T d B (same arithmetic and order as wino_input_kernel: bit-identical fp32 values) written as three bf16
// planes: v[plane][p][t][ci].  Thread = (tile, 4 channels): 16 coalesced 16-B loads, 48 8-B stores (512 B contiguous per wave and plane).
struct WinoInSplitArgs {
    const float* x;
    unsigned short* v;
    long long plane;       // elements per plane = 16 * T * Cin
    float scale;           // NP = 2: V is multiplied by this power of two before the fp16 split
    int B, H, W, Cin, ldx, th, tw, T, c4n;
    y2_fastdiv d_c4, d_tt, d_tw;
};

template <int NP>      // NP = 3: bf16 plane triples; NP = 2: fp16 plane pairs of V * a.scale
__global__ __launch_bounds__(256) void wino_input_split_kernel(const WinoInSplitArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = y2_div(idx, a.d_c4);
    if (t >= (uint32_t)a.T) return;
    const int c4 = (int)(idx - t * (uint32_t)a.c4n);
    const int b = (int)y2_div(t, a.d_tt);
    const int r = (int)t - b * a.th * a.tw;
    const int ty = (int)y2_div((uint32_t)r, a.d_tw);
    const int tx = r - ty * a.tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = y0 + i;
        const bool yok = (unsigned)yy < (unsigned)a.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = x0 + j;
            const bool ok = yok && (unsigned)xx < (unsigned)a.W;
            d[i][j] = ok ? *reinterpret_cast<const f32x4*>(a.x + ((size_t)(b * a.H + yy) * a.W + xx) * a.ldx + 4 * c4) : zero;
        }
    }
    f32x4 s[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = d[0][j] - d[2][j];
        s[1][j] = d[1][j] + d[2][j];
        s[2][j] = d[2][j] - d[1][j];
        s[3][j] = d[1][j] - d[3][j];
    }
    unsigned short* dst = a.v + (size_t)t * a.Cin + 4 * c4;
    const size_t pos = (size_t)a.T * a.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 o[4];
        o[0] = s[i][0] - s[i][2];
        o[1] = s[i][1] + s[i][2];
        o[2] = s[i][2] - s[i][1];
        o[3] = s[i][1] - s[i][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned aa, bb, cc = 0;
                if (NP == 3) split3(o[j][e], aa, bb, cc);
                else split2h(o[j][e] * a.scale, aa, bb);
                h[e] = (unsigned short)aa; m[e] = (unsigned short)bb; l[e] = (unsigned short)cc;
            }
            unsigned short* q = dst + (4 * i + j) * pos;
            *reinterpret_cast<u16x4*>(q) = h;
            *reinterpret_cast<u16x4*>(q + a.plane) = m;
            if (NP == 3) *reinterpret_cast<u16x4*>(q + 2 * a.plane) = l;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct GemmSplitArgs {
    const unsigned short* A;   // [3 planes][groups][M][K] bf16 bits
    const unsigned short* B;   // [3 planes][groups][N][K]
    float* C;                  // [groups][M][ldc]
    long long planeA, planeB;  // elements per plane
    long long gA, gB, gC;      // elements per group
    int M, N, K, ldc, tiles_m, tiles_n, groups;
    float out_scale;           // C = (acc) * out_scale (the fp16 variant's operands carry power-of-two scales)
    long long* stamps;         // debug builds (-DY2_STAMPS): cycle stamps of workgroup 0
    unsigned a_bytes, b_bytes; // bytes from a group's plane-0 slice to the end of its plane-2 slice (the buffer range; masked rows use an offset beyond it)
};

constexpr int GS_BM = 128, GS_BN = 128;

// BK = 32: 48 KB stages, 3-deep ring = 144 KB: one workgroup per CU (one wave per SIMD), 48 MFMAs per barrier.
// BK = 16: 24 KB stages, 3-deep ring = 72 KB: two workgroups per CU cover each other's barrier / LDS-read bubbles, 24 MFMAs per barrier.
// NW = 8 (512 threads, wave tile 64 x 32): two waves per SIMD - while one is held in an LDS-DMA issue (~60 cycles, twice a bf16 MFMA), in
// its fragment reads or at the barrier, the other feeds the matrix pipe; every wave issues half the DMA instructions of the 4-wave form.
// NP = 3: bf16 plane triples, six products per K step;  NP = 2: fp16 plane pairs (hi, lo), three products (hi x hi, hi x lo, lo x hi; the
// dropped lo x lo is 2^-24 of the product): half the MFMAs and 4 instead of 6 bytes per operand element - this kernel is power-limited
// (DESIGN.md 3.5), so fewer MFMAs and bytes are what buys speed.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NP> struct SplitOps;
template <> struct SplitOps<3> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag x, frag y, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); }
};
template <> struct SplitOps<2> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag x, frag y, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); }
};

template <int BK, int NW, int NP>
__global__ __launch_bounds__(NW * 64) void gemm_split_kernel(const GemmSplitArgs a) {
    typedef typename SplitOps<NP>::frag frag_t;
    constexpr int NPROD = NP == 3 ? 6 : 3;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int ROWB = BK * 2;                        // bytes per row and plane in a stage
    constexpr int LPR = ROWB / 16;                      // lanes (16-B chunks) per row
    constexpr int NT = NW * 64;
    constexpr int RPI = NT / LPR;                       // rows per DMA instruction of the workgroup
    constexpr int WN_WAVES = NW / 2;                    // wave grid 2 x (NW / 2)
    constexpr int WN = GS_BN / WN_WAVES;                // wave tile 64 x WN
    constexpr int MB = 2, NBK = WN / 32;                // 32 x 32 blocks per wave tile
    constexpr int PLANE_A = GS_BM * ROWB, PLANE_B = GS_BN * ROWB;
    constexpr int STAGE = NP * (PLANE_A + PLANE_B);
    constexpr int NA = GS_BM / RPI, NB = GS_BN / RPI;   // DMA instructions per plane
    constexpr int NDMA = NP * (NA + NB);                // ... per thread and slab
    constexpr int KSTEPS = BK / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
    const int l31 = lane & 31, half = lane >> 5;

    int tile = y2_xcd_remap(blockIdx.x, gridDim.x);
    const int tpg = a.tiles_m * a.tiles_n;
    const int grp = tile / tpg;
    tile -= grp * tpg;
    const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
    const int m0 = tile_m * GS_BM, n0 = tile_n * GS_BN;

    // ---- DMA assignment: one instruction = RPI rows x ROWB bytes; lane -> physical chunk pc = t % LPR of row t / LPR, it fetches the
    // logical chunk pc ^ swizzle(row); swizzle(row) = (row >> 2) & 3 for 64-B rows, (row >> 3) & 1 for 32-B rows (rows + RPI: the same)
    const int srow = t / LPR;
    const int swz = BK == 32 ? ((srow >> 2) & 3) : ((srow >> 3) & 1);
    const int lchunk = (t % LPR) ^ swz;
    unsigned a_off[NA], b_off[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + srow + RPI * i;
        a_off[i] = m < a.M ? (unsigned)(((size_t)m * a.K) * 2 + lchunk * 16) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + srow + RPI * i;
        b_off[i] = n < a.N ? (unsigned)(((size_t)n * a.K) * 2 + lchunk * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.A + (size_t)grp * a.gA), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.B + (size_t)grp * a.gB), 0, a.b_bytes, 0x00020000);
    const unsigned pa_bytes = (unsigned)(a.planeA * 2), pb_bytes = (unsigned)(a.planeB * 2);

    // piece j of a slab: j < 3 * NA: A plane j / NA, row block j % NA; then B likewise
    auto issue_piece = [&](int k0, int slot, int j) {
        unsigned char* base = gs_smem + slot * STAGE + wave * 1024;
        if (j < NP * NA) {
            const int p = j / NA, i = j % NA;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(base + p * PLANE_A + i * (NT * 16)), 16, (int)a_off[i], (int)(p * pa_bytes + (unsigned)k0 * 2u), 0, 0);
        } else {
            const int jj = j - NP * NA;
            const int p = jj / NB, i = jj % NB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(base + NP * PLANE_A + p * PLANE_B + i * (NT * 16)), 16, (int)b_off[i], (int)(p * pb_bytes + (unsigned)k0 * 2u), 0, 0);
        }
    };
    auto issue_slab = [&](int k0, int slot) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) issue_piece(k0, slot, j);
    };

    // two accumulator sets: `acc` takes the hi x hi products (K / 16 sequential additions), `accl` the five cross products, which are
    // 2^-8 ... 2^-16 of it: summed among themselves they keep their low bits, and they meet the main sum once, in the epilogue.  (One
    // accumulator for everything measured 1.3-2.3x the error of the fp32-MFMA kernels at K >= 512: every small product was rounded to
    // the big accumulator's last place.)
    f32x16 acc[MB][NBK], accl[MB][NBK];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accl[i][j][r] = 0.f; }

    // fragment byte offsets inside a plane: row (wave rows + l31 [+ 32 * block]), physical chunk of logical chunk 2*kk + half
    const int sw = BK == 32 ? ((l31 >> 2) & 3) : ((l31 >> 3) & 1);
    int foff[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) foff[kk] = l31 * ROWB + (((2 * kk + half) ^ sw) << 4);
    const int fa = wm * 64 * ROWB;
    const int fb = NP * PLANE_A + wn * WN * ROWB;

    // the six plane products of one K = 16 step in ascending magnitude: the accumulator meets the small terms first
    // (NP = 2: (lo, hi), (hi, lo), (hi, hi).)  The LAST product is hi x hi; the first NP entries name every plane once (fragment read order).
    constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
    constexpr int PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};

    const int nk = a.K / BK;
    // PF: the DMA of slab s+2 goes out piece by piece behind the first MFMAs of slab s (one piece per 2 MFMAs)
    auto compute_slab = [&](auto PF, int slot, int k_next, int slot_next) {
        constexpr bool prefetch = decltype(PF)::value;
        const unsigned char* sbuf = gs_smem + slot * STAGE;
        // all fragment reads of the slab go out first (those of the first K step in front): the MFMAs of step 0 run while step 1 lands
        frag_t af[KSTEPS][MB][NP], bfr[KSTEPS][NBK][NP];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
            for (int q = 0; q < NP; ++q) {         // planes in the order the products consume them (NP = 3: A 2, 0, 1 / B 0, 2, 1; NP = 2: A 1, 0 / B 0, 1)
                const int pa = PA[q], pb = PB[q];
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    af[kk][i][pa] = __builtin_bit_cast(frag_t, *reinterpret_cast<const f32x4*>(sbuf + fa + pa * PLANE_A + i * 32 * ROWB + foff[kk]));
#pragma unroll
                for (int j = 0; j < NBK; ++j)
                    bfr[kk][j][pb] = __builtin_bit_cast(frag_t, *reinterpret_cast<const f32x4*>(sbuf + fb + pb * PLANE_B + j * 32 * ROWB + foff[kk]));
            }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NMFMA = KSTEPS * NPROD * MB * NBK;
        constexpr int EVERY = (NMFMA / 2) / NDMA > 0 ? (NMFMA / 2) / NDMA : 1;      // one DMA piece per EVERY MFMAs over the first half of the slab
        int cnt = 0;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
            for (int q0 = 0; q0 < NPROD; ++q0)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NBK; ++j) {
                        const int q = q0;
                        if (q == NPROD - 1) acc[i][j] = SplitOps<NP>::mfma(af[kk][i][PA[q]], bfr[kk][j][PB[q]], acc[i][j]);
                        else accl[i][j] = SplitOps<NP>::mfma(af[kk][i][PA[q]], bfr[kk][j][PB[q]], accl[i][j]);
                        if (prefetch && (cnt % EVERY) == EVERY - 1 && (cnt / EVERY) < NDMA) {
                            issue_piece(k_next, slot_next, cnt / EVERY);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ++cnt;
                    }
#pragma unroll
        for (int j = NMFMA / EVERY; prefetch && j < NDMA; ++j) issue_piece(k_next, slot_next, j);
    };

#ifdef Y2_STAMPS
    // per-stage cycle stamps of wave 0 of workgroup 0 (s_memtime): [stage][0] before the DMA wait, [1] after it, [2] after the barrier, [3] after the MFMAs
    long long* stamps = (blockIdx.x == 0 && t == 0) ? a.stamps : nullptr;
#define Y2_STAMP(ks_, w_) do { if (stamps != nullptr && (ks_) < 64) stamps[(ks_) * 4 + (w_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define Y2_STAMP(ks_, w_) do { } while (0)
#endif
#ifdef Y2_STAMPS
    if (stamps != nullptr) { stamps[252] = (long long)__builtin_readcyclecounter(); stamps[253] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
    issue_slab(0, 0);
    if (nk > 1) issue_slab(BK, 1);
    int cur = 0, nxt = 2;
    // (the prefetching and the draining iterations are separate loops: one loop body with a branch would make the accumulators meet in
    // phi nodes and cost an accumulator copy per MFMA, see conv_wgrad.hip)
    for (int ks = 0; ks + 2 < nk; ++ks) {
        Y2_STAMP(ks, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");      // slab ks landed; slab ks+1 may still be in flight
        Y2_STAMP(ks, 1);
        __builtin_amdgcn_s_barrier();        // everyone's part of slab ks is in LDS; everyone finished reading slab ks-1 (whose slot is refilled now)
        Y2_STAMP(ks, 2);
        compute_slab(std::true_type{}, cur, (ks + 2) * BK, nxt);
        Y2_STAMP(ks, 3);
        cur = cur == 2 ? 0 : cur + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    if (nk >= 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        compute_slab(std::false_type{}, cur, 0, 0);
        cur = cur == 2 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute_slab(std::false_type{}, cur, 0, 0);

#ifdef Y2_STAMPS
    if (stamps != nullptr) { stamps[254] = (long long)__builtin_readcyclecounter(); stamps[255] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
    // ---- store: register r of block (i, j) is row 8*(r>>2) + 4*half + (r&3), column l31: a wave writes 128-B row segments
    float* C = a.C + (size_t)grp * a.gC;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                if (row < a.M && col < a.N) C[(size_t)row * a.ldc + col] = (acc[i][j][r] + accl[i][j][r]) * a.out_scale;
            }
        }
}

}  // namespace

extern "C" int y2_split_bf16x3(const float* src, void* dst, long long n, y2_stream_t stream) {
    if (src == nullptr || dst == nullptr || n <= 0) return Y2_EINVAL;
    if ((n & 3) != 0 || !y2_aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u) != 0) return Y2_EALIGN;
    const long long n4 = n / 4;
    long long grid = (n4 + 255) / 256;
    if (grid > (long long)Y2_NUM_CU * 16) grid = (long long)Y2_NUM_CU * 16;
    Y2_LAUNCH("split3_kernel", 0.0, split3_kernel, dim3((unsigned)grid), dim3(256), 0, y2_s(stream), src, static_cast<unsigned short*>(dst), n4, n);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// C[g] (M x N, row stride ldc, fp32) = A[g] (M x K) * B[g]^T (N x K) for `groups` problems; A / B are split planes (see GemmSplitArgs).
// Library-internal (wino.hip) and behind y2_gemm_split (tests, tools).
int y2_internal_gemm_split(const void* A, long long planeA, const void* B, long long planeB, float* C, long long M, int N, int K, int ldc, int groups, int planes,
                           float out_scale, y2_stream_t stream) {
    if (planes != 2 && planes != 3) return Y2_EINVAL;
    if (A == nullptr || B == nullptr || C == nullptr || M <= 0 || N <= 0 || K <= 0 || groups < 1 || ldc < N) return Y2_EINVAL;
    if ((K % 32) != 0) return Y2_ENOSUP;
    if (!y2_aligned16(A) || !y2_aligned16(B)) return Y2_EALIGN;
    // a group's three plane slices sit behind ONE buffer descriptor (plane offset = scalar offset): its range must stay below the 2^31 "masked" offset
    if (planeA <= 0) planeA = (long long)groups * M * K;
    if (planeB <= 0) planeB = (long long)groups * N * K;
    if (planeA < (long long)groups * M * K || planeB < (long long)groups * N * K || (planeA & 7) || (planeB & 7)) return Y2_EINVAL;
    const unsigned long long span_a = ((unsigned long long)(planes - 1) * planeA + (unsigned long long)M * K) * 2ull, span_b = ((unsigned long long)(planes - 1) * planeB + (unsigned long long)N * K) * 2ull;
    if (span_a >= 0x7fffffffull || span_b >= 0x7fffffffull) return Y2_ENOSUP;
    GemmSplitArgs a;
    a.A = static_cast<const unsigned short*>(A); a.B = static_cast<const unsigned short*>(B); a.C = C;
    a.gA = M * K; a.gB = (long long)N * K; a.gC = M * ldc;
    a.planeA = planeA; a.planeB = planeB;
    a.M = (int)M; a.N = N; a.K = K; a.ldc = ldc; a.groups = groups;
    a.tiles_m = y2_cdiv(M, GS_BM); a.tiles_n = y2_cdiv(N, GS_BN);
    a.a_bytes = (unsigned)span_a; a.b_bytes = (unsigned)span_b;
    const long long grid = (long long)a.tiles_m * a.tiles_n * groups;
    if (grid > 0x7fffffffLL || M > 0x7fffffffLL) return Y2_EINVAL;
    a.out_scale = out_scale;
    a.stamps = nullptr;
#ifdef Y2_STAMPS
    if (const char* sp = getenv("Y2_GS_STAMPS_PTR")) a.stamps = reinterpret_cast<long long*>(strtoull(sp, nullptr, 0));
#endif
    const char* e = getenv("Y2_SPLIT_BK");            // 16: 24 KB stages, two workgroups per CU; 32 (default): 48 KB stages, one (A/B runs)
    const int bk = (e != nullptr && atoi(e) == 16) ? 16 : 32;
    const char* e2 = getenv("Y2_SPLIT_WAVES");        // 4: one wave per SIMD (wave tile 64 x 64); 8 (default): two (64 x 32)
    const int nw = (e2 != nullptr && atoi(e2) == 4) ? 4 : 8;
    const char* name = planes == 2 ? (groups > 1 ? "gemm_split_f16_kernel[grouped]" : "gemm_split_f16_kernel") : (groups > 1 ? "gemm_split_kernel[grouped]" : "gemm_split_kernel");
    const double flops = 2.0 * (double)M * N * K * groups;
#define Y2_GS_LAUNCH(BK_, NW_, NP_)                                                                                                \
    do {                                                                                                                           \
        static Y2LdsAttr attr;                                                                                                     \
        if (const int rc = attr.ensure(reinterpret_cast<const void*>(gemm_split_kernel<BK_, NW_, NP_>))) return rc;                \
        Y2_LAUNCH(name, flops, (gemm_split_kernel<BK_, NW_, NP_>), dim3((unsigned)grid), dim3(NW_ * 64), (size_t)3 * NP_ * (GS_BM + GS_BN) * BK_ * 2, y2_s(stream), a); \
    } while (0)
    if (planes == 2) {
        if (nw == 4) Y2_GS_LAUNCH(32, 4, 2); else Y2_GS_LAUNCH(32, 8, 2);
    } else if (bk == 16) Y2_GS_LAUNCH(16, 4, 3);           // (32-B rows: one DMA instruction of 512 threads would span 256 rows)
    else if (nw == 4) Y2_GS_LAUNCH(32, 4, 3);
    else Y2_GS_LAUNCH(32, 8, 3);
#undef Y2_GS_LAUNCH
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_gemm_split(const void* A, const void* B, float* C, long long M, int32_t N, int32_t K, int32_t ldc, int32_t groups, y2_stream_t stream) {
    return y2_internal_gemm_split(A, 0, B, 0, C, M, N, K, ldc, groups, 3, 1.f, stream);
}

extern "C" int y2_gemm_split_f16(const void* A, const void* B, float* C, long long M, int32_t N, int32_t K, int32_t ldc, int32_t groups, float out_scale, y2_stream_t stream) {
    return y2_internal_gemm_split(A, 0, B, 0, C, M, N, K, ldc, groups, 2, out_scale, stream);
}

// 1 when any operand handed to the fp16 split (y2_split_f16x2, the input transform of Y2_ALGO_WINOGRAD_SPLIT_F16) exceeded fp16's range since the
// last reset - the results of those launches contain Inf / NaN; reset != 0 clears the flag.  Synchronises the device.
extern "C" int y2_split_f16_overflow(int reset) {
    int flag = 0;
    if (hipMemcpyFromSymbol(&flag, HIP_SYMBOL(y2_f16_overflow_flag), sizeof(int)) != hipSuccess) return Y2_EINVAL;
    if (reset) {
        const int zero = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(y2_f16_overflow_flag), &zero, sizeof(int)) != hipSuccess) return Y2_EINVAL;
    }
    return flag != 0 ? 1 : 0;
}

extern "C" int y2_split_f16x2(const float* src, void* dst, long long n, float scale, y2_stream_t stream) {
    if (src == nullptr || dst == nullptr || n <= 0 || !(scale > 0.f)) return Y2_EINVAL;
    if ((n & 3) != 0 || !y2_aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u) != 0) return Y2_EALIGN;
    const long long n4 = n / 4;
    long long grid = (n4 + 255) / 256;
    if (grid > (long long)Y2_NUM_CU * 16) grid = (long long)Y2_NUM_CU * 16;
    Y2_LAUNCH("split2h_kernel", 0.0, split2h_kernel, dim3((unsigned)grid), dim3(256), 0, y2_s(stream), src, static_cast<unsigned short*>(dst), n4, n, scale);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// Stage 1 of the Winograd path with split output (wino.hip): V planes [3][16][T][Cin] of the chunk's input.
int y2_internal_wino_input_split(const float* x, void* v, int B, int H, int W, int Cin, int ldx, int planes, float scale, y2_stream_t stream) {
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    if (x == nullptr || v == nullptr || (Cin % 4) != 0 || (ldx % 4) != 0 || T * (Cin / 4) >= 0xffffffffLL) return Y2_EINVAL;
    WinoInSplitArgs a;
    a.x = x; a.v = static_cast<unsigned short*>(v); a.plane = 16 * T * Cin;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.th = th; a.tw = tw; a.T = (int)T; a.c4n = Cin / 4;
    a.d_c4 = y2_make_fastdiv((uint32_t)a.c4n); a.d_tt = y2_make_fastdiv((uint32_t)(th * tw)); a.d_tw = y2_make_fastdiv((uint32_t)tw);
    a.scale = scale;
    if (planes == 2) Y2_LAUNCH("wino_input_split_kernel", 0.0, wino_input_split_kernel<2>, dim3((unsigned)y2_cdiv(T * a.c4n, 256)), dim3(256), 0, y2_s(stream), a);
    else Y2_LAUNCH("wino_input_split_kernel", 0.0, wino_input_split_kernel<3>, dim3((unsigned)y2_cdiv(T * a.c4n, 256)), dim3(256), 0, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
