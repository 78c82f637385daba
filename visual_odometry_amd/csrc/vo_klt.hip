// vo_klt.hip -- pyramidal Lucas-Kanade with forward-backward check, one wavefront per keypoint.
// Replaces vision.PointTracker as driven by KLTTracker.track (modules/KLTTracker.m:22-41; called
// at modules/VisualOdometry.m:71,278).  Normative algorithm: oracle/vo_oracle.h N3-N5.
//
// Mapping.  One 64-lane workgroup (= one wave) owns one keypoint for the forward and the backward
// pass.  The wu x wv window is cut into LU x LV rectangular blocks of BW x BH pixels, one block per
// lane, so that a lane's bilinear taps overlap: (BW+1)(BH+1) LDS reads serve BW*BH pixels (1.4
// reads/pixel for the 41x41 window instead of 4) and the interpolation is separable in registers.
// The template gradients (Tu, Tv) live in registers for a whole level; they are the Sobel/8
// operator applied to the sub-pixel-sampled template itself, so a frame's pyramid is ONE float
// plane per level.  The template intensities are not kept: the residual sum splits into
// b = sum T*g - sum J*g, and the first term is a per-level constant computed during set-up.
// The search patch of the other image is staged into LDS once per level (LDS-DMA, one patch row
// per wave instruction) with a margin of LK_M pixels and re-staged only when the estimate walks
// out of it: an iteration touches nothing but LDS and registers.  Patch pitches are chosen so that
// the 32-lane halves of a wave hit distinct banks (at most 2-way on a handful of lanes).  Wave
// sums use DPP row operations.
//
// Bound: FP32 VALU (+ LDS bandwidth); HBM traffic is the pyramids only (L2 / Infinity Cache hits).
// Algorithmic flops: set-up 24 flop/px, iteration 11 flop/px; ctx counters [0] and [2]
// count executed iterations and set-ups.
#include "vo_common.h"

namespace {

// LDS patch margin in pixels.  2: the 41 x 41 patch is 52 x 47 floats, 16 workgroups a CU (the 128-VGPR kernel's own
// limit); 3 made it 52 x 49 and 15 a CU; 4 (fewer re-stagings, 14 a CU) was 5 % slower
constexpr int LK_M = 2;

struct LKArgs {
    const float *A, *B;
    int nu0, nv0;
    int levels, wu, wv, npx, max_iters, n;
    float eps2;
    double max_bidir, flat_thr; // flat_thr = 64 * 2 * npx * VO_MIN_EIG: the flat test's bound on sums of the x8 gradients
    const double *pu, *pv;
    double *qu, *qv;
    uint8_t* valid;
    float *score, *fb;
    unsigned long long* counters; // VO_COUNTER_SHARDS x 4
    const int32_t* n_dev;         // optional: the launch covers an upper bound n, the number of keypoints is *n_dev on the device
    // head of the launch (vo_dev_klt_track_head): the first n_head keypoints are a set of their own whose consumers start
    // when the last of them is done, while the rest of the launch is still running.  head_cnt counts finished head
    // keypoints (zero between launches: the last arriver sets it back), head_flag receives head_epoch from the last arriver.
    int n_head; unsigned head_epoch; unsigned* head_cnt; unsigned* head_flag;
};

template <int BW_, int BH_, int LU_, int LV_, int PW_, bool COL_, int WAVES_>
struct Cfg {
    static constexpr int WAVES = WAVES_; // occupancy target (waves per SIMD) handed to the register allocator
    static constexpr int BW = BW_, BH = BH_, LU = LU_, LV = LV_, PW = PW_;
    static constexpr bool COL = COL_;
    static constexpr int PH = LV * BH + 1 + 2 * LK_M;
    static constexpr int MAXU = LU * BW, MAXV = LV * BH;
    static constexpr int NQ = PW / 4, RPI = 64 / NQ; // 16-byte pieces per patch row, rows per LDS-DMA instruction
    static_assert(PW % 4 == 0, "patch pitch must be a multiple of 4 (16-byte LDS-DMA rows)");
    static_assert(PW >= LU * BW + 1 + 2 * LK_M, "patch pitch too small");
    static_assert(LU * LV <= 64, "more blocks than lanes");
};
// pitches from a bank-conflict search over (lane -> block) orders, see DESIGN.md
using CfgS = Cfg<3, 3, 7, 7, 28, true, 5>;   // windows up to 21 x 21
using CfgM = Cfg<5, 4, 7, 8, 44, false, 4>;  // up to 35 x 32
using CfgL = Cfg<5, 6, 9, 7, 52, false, 4>;  // up to 45 x 42 (the reference's 41 x 41)
using CfgX = Cfg<7, 7, 8, 8, 68, true, 1>;   // up to 56 x 56 (51 x 51 of module_tests/updateW2D3D_test.m:20)

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xf, false));
}
// the same with bound_ctrl: a lane without a source lane reads 0 and the destination needs no prior value (one
// v_mov less per use than the form above, whose untouched lanes keep `old`)
template <int CTRL>
__device__ __forceinline__ float dpp_mov_z(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over the 64 lanes, result uniform
__device__ __forceinline__ float wave_sum(float x) {
    x += dpp_mov<0xb1, 0xf>(x);  // quad_perm [1,0,3,2]
    x += dpp_mov<0x4e, 0xf>(x);  // quad_perm [2,3,0,1]
    x += dpp_mov<0x114, 0xf>(x); // row_shr:4
    x += dpp_mov<0x118, 0xf>(x); // row_shr:8
    x += dpp_mov<0x142, 0xa>(x); // row_bcast:15
    x += dpp_mov<0x143, 0xc>(x); // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// ---- several wave sums at once.  A butterfly that halves the number of live values at every
// step: lanes with bit k of the lane index clear keep one value of a pair, lanes with it set the
// other, and each adds its partner lane's share.  After three steps lane l holds value (l & 7)
// summed over 8 lanes; one row rotation, one 16-lane and one 32-lane swap (gfx950 permlane swaps)
// complete the sums in every lane.  8 values cost 26 VALU + 8 readlanes instead of 8 x (6 + 1).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sel_lanes(float if_set, float if_clear, unsigned long long lanes) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(lanes));
    return r;
}
template <int CTRL>
__device__ __forceinline__ float merge_pair(float a, float b, unsigned long long lanes) {
    // lanes clear: a summed over the pair of lanes; lanes set: b summed over the pair
    return sel_lanes(b, a, lanes) + dpp_mov<CTRL, 0xf>(sel_lanes(a, b, lanes));
}
template <int N>
__device__ __forceinline__ void wave_sums(const float (&v)[N], float (&out)[N]) {
    static_assert(N >= 5 && N <= 8, "wave_sums handles 5..8 values");
    constexpr unsigned long long B0 = 0xaaaaaaaaaaaaaaaaull, B1 = 0xccccccccccccccccull, B2 = 0xf0f0f0f0f0f0f0f0ull;
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = 2 * k < N ? v[2 * k] : 0.0f, b = 2 * k + 1 < N ? v[2 * k + 1] : 0.0f;
        w[k] = 2 * k < N ? merge_pair<0xb1>(a, b, B0) : 0.0f;      // quad_perm [1,0,3,2]
    }
    const float x0 = merge_pair<0x4e>(w[0], w[1], B1);                 // quad_perm [2,3,0,1]
    const float x1 = merge_pair<0x4e>(w[2], w[3], B1);
    float y = merge_pair<0x124>(x0, x1, B2);                           // row_ror:4
    y += dpp_mov<0x128, 0xf>(y);                                       // row_ror:8: the whole row of 16
    const unsigned yb = (unsigned)__float_as_int(y);
    const u32x2 r16 = __builtin_amdgcn_permlane16_swap(yb, yb, false, false);
    const float z = __int_as_float((int)r16.x) + __int_as_float((int)r16.y);
    const unsigned zb = (unsigned)__float_as_int(z);
    const u32x2 r32 = __builtin_amdgcn_permlane32_swap(zb, zb, false, false);
    const float t = __int_as_float((int)r32.x) + __int_as_float((int)r32.y);
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), k));
}

// x where the lane's window-mask bit is set, +0.0f elsewhere: the bit is extracted sign-extended to an
// all-ones / all-zero word and ANDed in -- no per-pixel lane masks in SGPRs
__device__ __forceinline__ int mask_word(unsigned mlo, unsigned mhi, int bit) { // bit: a constant after unrolling
    return __builtin_amdgcn_sbfe((int)(bit < 32 ? mlo : mhi), (unsigned)(bit & 31), 1u); // v_bfe_i32: -1 or 0
}
// The window mask of a lane's block.  General form: one bit per pixel in a 64-bit word.  W41 form: the window is
// (LU-1) BW + 1 wide and LV BH - 1 high (41 x 41 on 9 x 7 blocks of 5 x 6, the reference's BlockSize): only the last
// column of blocks (columns 1.. outside) and the last row of blocks (last row outside) are cut, so four whole-word
// masks -- lane alive / not in the last column of blocks / not in the last row of blocks / neither -- replace the
// per-pixel bit extraction.
template <bool W41> struct WinMask;
template <> struct WinMask<false> {
    unsigned lo, hi;
    __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(lo), "+v"(hi)); }
    template <int BW, int BH> __device__ __forceinline__ int word(int i, int j) const { return mask_word(lo, hi, j * BW + i); }
};
template <> struct WinMask<true> {
    int a, c, r, cr;
    __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(a), "+v"(c), "+v"(r), "+v"(cr)); }
    template <int BW, int BH> __device__ __forceinline__ int word(int i, int j) const {
        return i >= 1 ? (j == BH - 1 ? cr : c) : (j == BH - 1 ? r : a);
    }
};
__device__ __forceinline__ float and_bits(float x, int m) { return __int_as_float(__float_as_int(x) & m); }

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// pyramid geometry of level l (wave-uniform scalar arithmetic; same rule as frame_prepare): every level is stored
// with a replicated border of VO_PYR_PAD pixels (vo_common.h); org = offset of the level's pixel (0, 0)
constexpr int PYR_PAD = VO_PYR_PAD;
__device__ __forceinline__ void level_dims(int nu0, int nv0, int l, int& nu, int& nv, long& org, int& pitch) {
    nu = nu0; nv = nv0;
    long off = 0;
    for (int k = 0; k < l; ++k) {
        off += (long)(nu + 2 * PYR_PAD) * (nv + 2 * PYR_PAD);
        nu = (nu + 1) >> 1; nv = (nv + 1) >> 1;
    }
    pitch = nu + 2 * PYR_PAD;
    org = off + (long)PYR_PAD * pitch + PYR_PAD;
}

// Stage the whole PW x PH patch (replicate-clamped coordinates): every LDS word a lane can read
// holds image data, so masked pixels never see NaN garbage.  Lane = patch column; the row walks a
// wave-uniform loop, so the row base is scalar and each element costs one saddr global load and one
// ds_write with an immediate offset -- no per-element address arithmetic on the VALU.
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(1))) const float glb_float;

// stage_issue starts the transfers (the patch must not be read, and not be waited for twice, until stage_wait has
// run); stage = the two together.  Split so that the first search patch of a level can travel while the level's
// template sums and 2x2 system are being finished (lk_dir).
template <class C>
__device__ __forceinline__ void stage_issue(float* patch, const float* __restrict__ img, int pitch, int nu, int nv, int ou, int ov,
                                            int lane, int lr, int lq) {
#if defined(__HIP_DEVICE_COMPILE__) // the host pass cannot type-check the 16-byte LDS-DMA builtin
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a patch sent ahead that nobody waited for (flat level) has landed
    __syncthreads();
    lds_float* lp = reinterpret_cast<lds_float*>(reinterpret_cast<uintptr_t>(patch));
    // A caller's keypoint may be anything (infinite, 1e12: the float -> int conversion saturates and the
    // window offsets then wrap).  Beyond +-2^24 every patch is the replicated border anyway, so the origin
    // is pinned there (scalar min / max) and none of the sums below can overflow.
    constexpr int FAR = 1 << 24;
    ou = clampi(ou, -FAR, FAR);
    ov = clampi(ov, -FAR, FAR);
    if (ou >= -PYR_PAD && ou + C::PW <= nu + PYR_PAD && ov >= -PYR_PAD && ov + C::PH <= nv + PYR_PAD) {
        // The patch lies inside the stored plane (image + replicated border; wave-uniform test, true for every
        // patch around a position inside the image unless the window is wider than 2 PAD - 7): 16-byte LDS-DMA
        // with nothing clamped.  The LDS image is lane-linear per instruction (base + lane * 16), so with PW a
        // multiple of 4 lanes (lr, lq) = (lane / NQ, lane % NQ) fill RPI whole rows per instruction from per-lane
        // source addresses that need only 4-byte alignment (scripts/ubench/glds16.hip); the address just advances.
        constexpr int NQ = C::NQ, RPI = C::RPI, NI = (C::PH + RPI - 1) / RPI;
        // Addresses: scalar base (the patch's first pixel, advanced RPI rows per instruction on the scalar unit) + one
        // 32-bit per-lane offset that depends on the level's pitch only -- the instruction's saddr form, written out
        // because the builtin takes one per-lane 64-bit pointer (a 64-bit vector add per instruction, a dozen vector
        // instructions of address set-up per patch, 17 patches a keypoint: 3 % of the kernel's vector instructions).
        // M0 = LDS byte address of the instruction's first row (one wait state before the transfer reads it).
        if (lane < RPI * NQ) {
            const float* sb = img + ((long)ov * pitch + ou);
            const unsigned voff = (unsigned)(lr * pitch + 4 * lq) * 4u;
            const unsigned l0 = (unsigned)reinterpret_cast<uintptr_t>(patch);
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                if ((k + 1) * RPI <= C::PH || lr < C::PH - k * RPI)
                    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(voff), "s"(sb), "s"(l0), "i"(k * RPI * C::PW * 4) : "memory", "m0", "scc");
                sb += RPI * pitch;
            }
        }
    } else {
        // anywhere else (positions outside the image, very wide windows at a border): 4-byte LDS-DMA, lane = patch
        // column, rows and columns replicate-clamped
#pragma unroll
        for (int c0 = 0; c0 < C::PW; c0 += 64) {
            const int c = c0 + lane;
            if (c < C::PW) {
                const int cc = clampi(ou + c, 0, nu - 1);
#pragma unroll 1
                for (int r = 0; r < C::PH; ++r) {
                    const float* rowp = img + (long)clampi(ov + r, 0, nv - 1) * pitch;
                    __builtin_amdgcn_global_load_lds((glb_float*)(rowp + cc), lp + r * C::PW + c0, 4, 0, 0);
                }
            }
        }
    }
#endif
}
__device__ __forceinline__ void stage_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}
template <class C>
__device__ __forceinline__ void stage(float* patch, const float* __restrict__ img, int pitch, int nu, int nv, int ou, int ov,
                                      int lane, int lr, int lq) {
    stage_issue<C>(patch, img, pitch, nu, nv, ou, ov, lane, lr, lq);
    stage_wait();
}

// -DVO_KLT_MARKS: comment lines in the assembly at the borders of the kernel's sections (scripts/klt_budget.py counts the
// instructions between them); nothing in a normal build
#ifdef VO_KLT_MARKS
#define LK_MARK(name) asm volatile("; LKMARK " name ::: "memory")
#else
#define LK_MARK(name)
#endif
__device__ __forceinline__ float unif(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// one direction A -> B (N4).  returns flags: bit0 lost, bit1 flat at level 0.
// Wave-uniform scalars (sub-pixel fractions, the 2x2 inverse, the running displacement) are pushed
// back to SGPRs after every VALU step so that the register file holds little besides the template.
template <class C, bool SCORE, bool W41>
__device__ int lk_dir(const LKArgs& a, const float* __restrict__ A, const float* __restrict__ B, double su, double sv,
                      float* patch, int lane, int lr, int lq, int lbase, const WinMask<W41>& mask, double& qu, double& qv, float& ssd_out,
                      unsigned& iters, unsigned& setups, unsigned& cells) {
    constexpr int BW = C::BW, BH = C::BH, PW = C::PW;
    // The residual sum is split, b = sum T*(Tu,Tv) - sum J*(Tu,Tv): the first term is a constant of the
    // level, so the template intensities T never live in registers -- only the gradients do.
    float Tu[BH][BW], Tv[BH][BW];
    const int ru = (a.wu - 1) / 2, rv = (a.wv - 1) / 2;
    const int nu0 = a.nu0, nv0 = a.nv0;
    float gu = 0, gv = 0;   // displacement carried from level to level: tens of pixels at most, fp32 keeps 1e-6 px of it
    int flags = 0;
    // A start that is not finite or is absurdly far away is lost before anything is staged: every path below
    // would end there too (flat at every level or the first lost test, displacement 0), and from here on all
    // positions are bounded -- the start by this test, each step by the lost test before a B patch is staged.
    if (!(fabs(su) <= 4194304.0 && fabs(sv) <= 4194304.0)) {
        qu = su; qv = sv; ssd_out = 0.0f;
        return 1;
    }
    bool staged0 = false; // level 0: is a B patch staged, and where (re-used by the score)
    int pou0 = 0, pov0 = 0;
    // The start as integer pixel + fraction: a level's position su / 2^l is then a shift and a mask on the integer and one
    // fp32 scaling of the remainder, and the per-level position arithmetic needs no fp64 (eight levels x directions of ~20
    // double-rate instructions on wave-uniform values; the fraction's rounding is below 3e-7 px).
    const double fsu_d = floor(su), fsv_d = floor(sv);
    const int isu = uni((int)fsu_d), isv = uni((int)fsv_d);
    const float fsu = unif((float)(su - fsu_d)), fsv = unif((float)(sv - fsv_d));
    for (int l = a.levels - 1; l >= 0; --l) {
        LK_MARK("level");
        int nu, nv, pitch; long lo;
        level_dims(nu0, nv0, l, nu, nv, lo, pitch);
        const int lmask = (1 << l) - 1;
        const int iplu = isu >> l, iplv = isv >> l;                      // floor(su / 2^l), negative starts included
        const float fplu = unif(__builtin_ldexpf((float)(isu & lmask) + fsu, -l));
        const float fplv = unif(__builtin_ldexpf((float)(isv & lmask) + fsv, -l));
        float sxx = 0, sxy = 0, syy = 0, ctu = 0, ctv = 0;
        // where the iteration of this level starts: position = (ibu + fbu + du, ibv + fbv + dv), integer base + float
        // remainder.  Known before the set-up, so the first search patch is sent for as soon as the template's samples
        // have left the LDS, and travels while the sums and the 2x2 system are being finished.
        const float bu0 = fplu + gu, bv0 = fplv + gv;
        const int ibu = iplu + uni((int)floorf(bu0)), ibv = iplv + uni((int)floorf(bv0));
        const float fbu = unif(__builtin_amdgcn_fractf(bu0)), fbv = unif(__builtin_amdgcn_fractf(bv0));
        int pou = ibu - ru - LK_M, pov = ibv - rv - LK_M;
        {   // ---- template set-up: E = samples on the window grown by one pixel, Sobel/8 of E.
            // Rows stream through a 3-row ring so that only 3 x (BW+2) samples are live at a time.
            const float fu = fplu, fv = fplv;
            LK_MARK("stageA");
            stage<C>(patch, A + lo, pitch, nu, nv, iplu - ru - 1, iplv - rv - 1, lane, lr, lq);
            // E rows stream through a 3-row ring.  Separable Sobel per template row: VD = vertical
            // difference, VS = vertical 1-2-1 smoothing of the three rows, then the horizontal halves.
            // Gradients are kept times 8 (the 1/8 of the operator is a power of two: it is folded,
            // exactly, into G and the 2x2 inverse below).
            // opaque per-level copies: the mask words are re-extracted at each use (one v_bfe_i32) instead
            // of living in 30 registers across the kernel
            LK_MARK("rows");
            WinMask<W41> wm = mask;
            wm.opaque();
            // Row-major lane order (lane + 1 = the block to the right): a lane samples only the BW columns that
            // start its block.  What completes its Sobel support -- the vertical difference and the vertical
            // smoothing of the first two columns of its right neighbour, and that neighbour's first centre sample --
            // is the same arithmetic on the same samples in lane + 1 and comes from there (DPP wave shift): 5
            // instead of 7 columns of separable lerps AND of vertical Sobel halves per row, 6 instead of 8 LDS reads.
            // The last block of a row of blocks has no such neighbour (lane + 1 is the first block of the next row):
            // what it receives must only reach masked pixels, which holds when the window leaves that block at most
            // BW - 2 columns -- the W41 window shape leaves it one (35 x 31 on the same blocks fills it: found by
            // test_klt_window_classes_and_borders).  Other shapes sample all BW + 2 columns themselves.
            constexpr bool SHARE = W41 && !C::COL && BW >= 3;
            constexpr int EC = SHARE ? BW : BW + 2;   // columns of E this lane samples itself
            float E[3][EC];
            float hp[EC];
#pragma unroll
            for (int r = 0; r < BH + 3; ++r) {
                float p[EC + 1];
#pragma unroll
                for (int c = 0; c < EC + 1; ++c) p[c] = patch[lbase + r * PW + c];
#pragma unroll
                for (int c = 0; c < EC; ++c) p[c] = p[c] + fu * (p[c + 1] - p[c]); // p[c] := H(r, c)
                if (r > 0) {
                    float* er = E[(r - 1) % 3];
#pragma unroll
                    for (int c = 0; c < EC; ++c) er[c] = hp[c] + fv * (p[c] - hp[c]);
                }
#pragma unroll
                for (int c = 0; c < EC; ++c) hp[c] = p[c];
                if (r >= 3) { // E rows j, j+1, j+2 are complete: emit template row j
                    const int j = r - 3;
                    const float* e0 = E[j % 3]; const float* e1 = E[(j + 1) % 3]; const float* e2 = E[(j + 2) % 3];
                    float VD[BW + 2], VS[BW + 2];
#pragma unroll
                    for (int c = 0; c < EC; ++c) {
                        VD[c] = e2[c] - e0[c];
                        VS[c] = (e0[c] + 2.0f * e1[c]) + e2[c];
                    }
                    float ectr;   // the centre row's sample in column BW (the template intensity of pixel BW - 1)
                    if constexpr (SHARE) {
                        VD[BW] = dpp_mov_z<0x130>(VD[0]);       // wave_shl:1 -- lane i reads lane i + 1
                        VD[BW + 1] = dpp_mov_z<0x130>(VD[1]);
                        VS[BW] = dpp_mov_z<0x130>(VS[0]);
                        VS[BW + 1] = dpp_mov_z<0x130>(VS[1]);
                        ectr = dpp_mov_z<0x130>(e1[0]);
                    } else {
                        ectr = e1[BW];
                    }
#pragma unroll
                    for (int i = 0; i < BW; ++i) {
                        const int in = wm.template word<BW, BH>(i, j);
                        const float tu = VS[i + 2] - VS[i];
                        const float tv = (VD[i] + 2.0f * VD[i + 1]) + VD[i + 2];
                        Tu[j][i] = and_bits(tu, in);
                        Tv[j][i] = and_bits(tv, in);
                        sxx += Tu[j][i] * Tu[j][i]; sxy += Tu[j][i] * Tv[j][i]; syy += Tv[j][i] * Tv[j][i];
                        const float ti = i + 1 < BW ? e1[i + 1] : ectr;
                        ctu += ti * Tu[j][i]; ctv += ti * Tv[j][i];
                    }
                }
            }
            ++setups;
            // every lane has its template samples in registers: the LDS is free for the search patch (a start far
            // outside the image goes through the clamped form inside; the lost test of the first iteration still
            // comes before anything is read from it)
            // (the five sums are pinned in front of the transfers -- an empty volatile asm that reads them: left free, the
            // compiler hoisted all of the rows' LDS reads in front of the transfers' barrier and sank their arithmetic
            // behind it, 54 samples live at once, and the set-up ran 2 % longer)
            asm volatile("" :: "v"(sxx), "v"(sxy), "v"(syy), "v"(ctu), "v"(ctv));
            LK_MARK("stageB");
            stage_issue<C>(patch, B + lo, pitch, nu, nv, pou, pov, lane, lr, lq);
        }
        // sums of the x8 gradients: S = 64 G
        LK_MARK("sums");
        float s5in[5] = {sxx, sxy, syy, ctu, ctv}, s5[5];
        wave_sums<5>(s5in, s5);
        LK_MARK("solve");
        const float CTu = s5[3], CTv = s5[4];
        // Flat test mineig(G) / npx >= VO_MIN_EIG without the eigenvalue: with t = 64 * 2 npx VO_MIN_EIG (LKArgs::flat_thr)
        //   2 mineig(S) >= t  <=>  (2 Suu - t) + (2 Svv - t) >= 0  and  (2 Suu - t)(2 Svv - t) >= (2 Suv)^2
        // (the characteristic polynomial of S at t / 2 is non-negative left of both eigenvalues).  A handful of fp64
        // operations on wave-uniform values -- the product difference cancels for edge-like windows, which is why it is
        // not fp32 -- instead of an fp64 square root and an fp64 division (~45 double-rate instructions per set-up,
        // eight set-ups a keypoint; an fp64 instruction costs the vector unit 1.6 fp32 ones).
        const double Suu = (double)s5[0], Suv = (double)s5[1], Svv = (double)s5[2];
        const double ea = fma(2.0, Suu, -a.flat_thr), eb = fma(2.0, Svv, -a.flat_thr);
        const double suv2 = Suv * Suv;
        const bool flat = !((ea + eb >= 0.0) && (fma(ea, eb, -4.0 * suv2) >= 0.0));
        float du = 0, dv = 0;
        if (flat) {
            if (l == 0) flags |= 2;
        } else {
            // inv(G) / 8 (the residual sums carry the x8 of the gradients) = 8 adj(S) / det(S): the determinant in fp64 (it
            // cancels too), its reciprocal in fp32 with one Newton step (the update needs no more: the fixed point of the
            // iteration does not depend on the inverse, only the path to it)
            const float detS = (float)fma(Suu, Svv, -suv2);
            float rdet = __builtin_amdgcn_rcpf(detS);
            rdet = rdet * fmaf(-detS, rdet, 2.0f);
            const float r8 = 8.0f * rdet;
            const float iuu = unif(s5[2] * r8), iuv = unif(-s5[1] * r8), ivv = unif(s5[0] * r8);
            // N4 lost test, (ibu + tu) 2^l inside [0, nu0 - 1], as bounds on tu itself (2^l is a power of two and the base
            // an integer: the bounds are exact)
            const float isf = __builtin_ldexpf(1.0f, -l);
            const float tlo_u = -(float)ibu, thi_u = (float)(nu0 - 1) * isf - (float)ibu;
            const float tlo_v = -(float)ibv, thi_v = (float)(nv0 - 1) * isf - (float)ibv;
            bool staged = true, landed = false;   // the patch sent for above: staged, not waited for yet
            // sum J*(Tu,Tv) is bilinear in the sub-pixel fractions: inside one pixel cell it is the
            // interpolation of the four integer-offset correlations c[dv][du] = sum P(x+du, y+dv)*(Tu,Tv).
            // They are computed once per visited cell; an iteration inside the cell is scalar work.
            // residual inside the cell: B = k0 - fu*ka - fv*(kb + fu*kc), per gradient
            float ku0 = 0, kua = 0, kub = 0, kuc = 0, kv0 = 0, kva = 0, kvb = 0, kvc = 0;
            // The cell whose correlations are in hand is remembered as the FLOATS floor(tu), floor(tv) (cru, crv; 0.5: none
            // yet -- no floor equals it), and the test an iteration makes is ONE: "still in the cell I tested last" (ctu, ctv:
            // the same floors if that cell lies wholly inside the image, else NaN, which no floor equals -- a cell that the
            // image border cuts is tested every time).  Outside the image and another cell are told apart behind that
            // branch.  An iteration that stays in its cell -- three of four -- is two additions, two floors, two compares and
            // one branch in front of the update, with no integer conversion and no trip through the scalar registers (the
            // loop was 65 instructions a trip, half of them scalar bookkeeping around three exits; a wave's time is its
            // instruction count, whatever unit executes them).
            float cru = 0.5f, crv = 0.5f, ctu = __builtin_nanf(""), ctv = __builtin_nanf("");
            int it = 0;
            if (a.max_iters > 0) for (;;) {
                LK_MARK("iter");
                const float tu = fbu + du, tv = fbv + dv;
                const float ftu = floorf(tu), ftv = floorf(tv);
                if (__builtin_amdgcn_ballot_w64((ftu != ctu) | (ftv != ctv))) {
                    // N4 lost test (bitwise &: the compares feed one scalar mask, no branch between them)
                    if (__builtin_amdgcn_ballot_w64(!((tu >= tlo_u) & (tu <= thi_u) & (tv >= tlo_v) & (tv <= thi_v)))) {
                        flags |= 1;
                        break;
                    }
                    if (__builtin_amdgcn_ballot_w64((ftu != cru) | (ftv != crv))) {
                        const int iu = ibu + uni((int)ftu), iv = ibv + uni((int)ftv);
                        LK_MARK("cell");
                        int fi = iu - ru - pou, fj = iv - rv - pov;
                        if (!staged || fi < 0 || fi > 2 * LK_M || fj < 0 || fj > 2 * LK_M) {
                            pou = iu - ru - LK_M; pov = iv - rv - LK_M;
                            stage<C>(patch, B + lo, pitch, nu, nv, pou, pov, lane, lr, lq);
                            staged = true; landed = true;
                            fi = LK_M; fj = LK_M;
                        }
                        if (!landed) { stage_wait(); landed = true; }
                        const int base = lbase + fj * PW + fi;
                        float su00 = 0, su10 = 0, su01 = 0, su11 = 0, sv00 = 0, sv10 = 0, sv01 = 0, sv11 = 0;
#pragma unroll
                        for (int r = 0; r < BH + 1; ++r) {
                            float p[BW + 1];
#pragma unroll
                            for (int c = 0; c < BW + 1; ++c) p[c] = patch[base + r * PW + c];
                            if (r < BH) {
#pragma unroll
                                for (int c = 0; c < BW; ++c) {
                                    su00 += p[c] * Tu[r][c]; su10 += p[c + 1] * Tu[r][c];
                                    sv00 += p[c] * Tv[r][c]; sv10 += p[c + 1] * Tv[r][c];
                                }
                            }
                            if (r > 0) {
#pragma unroll
                                for (int c = 0; c < BW; ++c) {
                                    su01 += p[c] * Tu[r - 1][c]; su11 += p[c + 1] * Tu[r - 1][c];
                                    sv01 += p[c] * Tv[r - 1][c]; sv11 += p[c + 1] * Tv[r - 1][c];
                                }
                            }
                        }
                        LK_MARK("cellsums");
                        float c8in[8] = {su00, su10, su01, su11, sv00, sv10, sv01, sv11}, c8[8];
                        wave_sums<8>(c8in, c8);
                        const float cu00 = c8[0], cu10 = c8[1], cu01 = c8[2], cu11 = c8[3];
                        const float cv00 = c8[4], cv10 = c8[5], cv01 = c8[6], cv11 = c8[7];
                        ku0 = CTu - cu00; kua = cu10 - cu00; kub = cu01 - cu00; kuc = (cu11 - cu10) - kub;
                        kv0 = CTv - cv00; kva = cv10 - cv00; kvb = cv01 - cv00; kvc = (cv11 - cv10) - kvb;
                        cru = ftu; crv = ftv;
                        ++cells;
                        // may the next iterations skip the lost test?  Only if no position of this cell is outside
                        const bool inside = (ftu >= tlo_u) & (ftu + 1.0f <= thi_u) & (ftv >= tlo_v) & (ftv + 1.0f <= thi_v);
                        ctu = inside ? ftu : __builtin_nanf(""); ctv = ftv;
                    }
                }
                LK_MARK("update");
                const float fu = __builtin_amdgcn_fractf(tu), fv = __builtin_amdgcn_fractf(tv);   // = tu - floor(tu)
                const float Bu = (ku0 - fu * kua) - fv * (kub + fu * kuc);
                const float Bv = (kv0 - fu * kva) - fv * (kvb + fu * kvc);
                const float ddu = fmaf(iuu, Bu, iuv * Bv);
                const float ddv = fmaf(iuv, Bu, ivv * Bv);
                du += ddu; dv += ddv;
                ++it;
                // one exit test for "converged" and "out of iterations": the last iteration compares with +inf (a step that is
                // infinite or NaN there goes round once more and leaves through the lost test, as it would a level later)
                const float lim = it >= a.max_iters ? __builtin_inff() : a.eps2;
                if (__builtin_amdgcn_ballot_w64(fmaf(ddu, ddu, ddv * ddv) < lim)) break;
            }
            iters += it;   // updates executed (an iteration that ends in the lost test is not one)
            if (l == 0) { staged0 = staged; pou0 = pou; pov0 = pov; }
        }
        LK_MARK("levelend");
        if (flags & 1) break;
        if (l > 0) { gu = 2.0f * (gu + du); gv = 2.0f * (gv + dv); }
        else { gu = gu + du; gv = gv + dv; }
    }
    // a search patch sent ahead that nothing waited for (a flat level, a point lost at its first iteration) must have
    // landed before the LDS is used again -- by this wave, or by the workgroup that gets the LDS after this one ends
    LK_MARK("dirend");
    stage_wait();
    qu = su + (double)gu; qv = sv + (double)gv;
    if (!(flags & 1)) {
        if (!(qu >= 0.0 && qu <= (double)(nu0 - 1) && qv >= 0.0 && qv <= (double)(nv0 - 1))) flags |= 1;
    }
    ssd_out = 0.0f;
    if (SCORE && !(flags & 1)) { // N5 residual at level 0: the match J first (the level-0 B patch is
                                 // usually still staged around it), then the re-sampled template
        float J[BH][BW];
        constexpr int pitch0_pad = 2 * PYR_PAD;
        const int pitch0 = nu0 + pitch0_pad;
        const long lo0 = (long)PYR_PAD * pitch0 + PYR_PAD;   // level 0's pixel (0, 0)
        {
            const double flu = floor(qu), flv = floor(qv);
            const int iu = uni((int)flu), iv = uni((int)flv);
            int fi = iu - ru - pou0, fj = iv - rv - pov0;
            if (!staged0 || fi < 0 || fi > 2 * LK_M || fj < 0 || fj > 2 * LK_M) {
                stage<C>(patch, B + lo0, pitch0, nu0, nv0, iu - ru - LK_M, iv - rv - LK_M, lane, lr, lq);
                fi = LK_M; fj = LK_M;
            }
            const float fu = unif((float)(qu - flu)), fv = unif((float)(qv - flv));
            const int base = lbase + fj * PW + fi;
            float hp[BW];
#pragma unroll
            for (int r = 0; r < BH + 1; ++r) {
                float p[BW + 1];
#pragma unroll
                for (int c = 0; c < BW + 1; ++c) p[c] = patch[base + r * PW + c];
#pragma unroll
                for (int c = 0; c < BW; ++c) p[c] = p[c] + fu * (p[c + 1] - p[c]);
                if (r > 0) {
#pragma unroll
                    for (int c = 0; c < BW; ++c) J[r - 1][c] = hp[c] + fv * (p[c] - hp[c]);
                }
#pragma unroll
                for (int c = 0; c < BW; ++c) hp[c] = p[c];
            }
        }
        const double flu = floor(su), flv = floor(sv);
        stage<C>(patch, A + lo0, pitch0, nu0, nv0, uni((int)flu) - ru - LK_M, uni((int)flv) - rv - LK_M, lane, lr, lq);
        const float fu = unif((float)(su - flu)), fv = unif((float)(sv - flv));
        const int base = lbase + LK_M * PW + LK_M;
        float ssd = 0;
        float hp[BW];
        WinMask<W41> wm = mask;
        wm.opaque();
#pragma unroll
        for (int r = 0; r < BH + 1; ++r) {
            float p[BW + 1];
#pragma unroll
            for (int c = 0; c < BW + 1; ++c) p[c] = patch[base + r * PW + c];
#pragma unroll
            for (int c = 0; c < BW; ++c) p[c] = p[c] + fu * (p[c + 1] - p[c]);
            if (r > 0) {
#pragma unroll
                for (int c = 0; c < BW; ++c) {
                    const int in = wm.template word<BW, BH>(c, r - 1);
                    const float e = and_bits((hp[c] + fv * (p[c] - hp[c])) - J[r - 1][c], in);
                    ssd += e * e;
                }
            }
#pragma unroll
            for (int c = 0; c < BW; ++c) hp[c] = p[c];
        }
        ssd_out = wave_sum(ssd);
    }
    return flags;
}

// SCORE: the N5 match score is computed (two more patch stagings and a window pass per keypoint); without it -- the
// caller passed no score array: modules/VisualOdometry.m:71,278 discard the third output of track -- it is skipped
template <class C, bool W41, bool SCORE>
__global__ __launch_bounds__(64, C::WAVES) void k_lk(LKArgs a) {
    __shared__ __attribute__((aligned(16))) float patch[C::PW * C::PH];
    const int lane = threadIdx.x, kp = blockIdx.x;
    if (a.n_dev && kp >= *a.n_dev) return;   // uniform
    const int lr = lane / C::NQ, lq = lane - lr * C::NQ; // this lane's (row, 16-byte piece) in a staging instruction
    // lane -> block; lanes beyond LU*LV read block 0's pixels with an all-masked window
    int lu, lv;
    const bool live = lane < C::LU * C::LV;
    if (C::COL) { lu = lane / C::LV; lv = lane - lu * C::LV; } else { lv = lane / C::LU; lu = lane - lv * C::LU; }
    const int lbase = live ? (lv * C::BH) * C::PW + lu * C::BW : 0;
    // bit (j * BW + i) = pixel (i, j) of this lane's block lies inside the wu x wv window
    static_assert(C::BW * C::BH <= 64, "block mask does not fit");
    WinMask<W41> wmask;
    if constexpr (W41) {
        wmask.a = live ? -1 : 0;
        wmask.c = (live && lu < C::LU - 1) ? -1 : 0;
        wmask.r = (live && lv < C::LV - 1) ? -1 : 0;
        wmask.cr = wmask.c & wmask.r;
    } else {
        unsigned long long mask64 = 0;
        if (live) {
#pragma unroll
            for (int j = 0; j < C::BH; ++j)
#pragma unroll
                for (int i = 0; i < C::BW; ++i)
                    if (lu * C::BW + i < a.wu && lv * C::BH + j < a.wv) mask64 |= 1ull << (j * C::BW + i);
        }
        wmask.lo = (unsigned)mask64; wmask.hi = (unsigned)(mask64 >> 32);
    }
    const double p0u = a.pu[kp], p0v = a.pv[kp];
    unsigned iters = 0, setups = 0, cells = 0;
    double fu, fv, bu = 0, bv = 0;
    float ssd, ssd2;
    const int fl = lk_dir<C, SCORE, W41>(a, a.A, a.B, p0u, p0v, patch, lane, lr, lq, lbase, wmask, fu, fv, ssd, iters, setups, cells);
    bool ok = (fl == 0);
    float fb = __builtin_nanf("");
    if (!(fl & 1) && a.max_bidir >= 0.0) {
        const int fl2 = lk_dir<C, false, W41>(a, a.B, a.A, fu, fv, patch, lane, lr, lq, lbase, wmask, bu, bv, ssd2, iters, setups, cells);
        if (fl2 & 1) ok = false;
        else {
            const double eu = p0u - bu, ev = p0v - bv;
            const double d = sqrt(fma(eu, eu, ev * ev));
            fb = (float)d;
            if (!(d <= a.max_bidir)) ok = false;
        }
    }
    if (lane == 0) {
        if (kp < a.n_head) {
            // A head keypoint: its results are read by kernels that start while this launch is still running (other
            // CUs, other XCDs, no kernel boundary behind this launch yet).  Write-through stores at agent scope, drained,
            // then the arrival count (MI355X_MICROARCH.md, "Valid forms": every store sc1, every storing wave's vmcnt(0)
            // before its add; the readers are later kernels and acquire at their own start).
            __hip_atomic_store(&a.qu[kp], fu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.qv[kp], fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.valid[kp], (uint8_t)(ok ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned before = __hip_atomic_fetch_add(a.head_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned n_head = a.n_dev ? (unsigned)min(a.n_head, *a.n_dev) : (unsigned)a.n_head;
            if (before + 1u == n_head) {
                __hip_atomic_store(a.head_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.head_flag, a.head_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            a.qu[kp] = fu; a.qv[kp] = fv;
            a.valid[kp] = ok ? 1 : 0;
        }
        if constexpr (SCORE) {
            float sc = 0.0f;
            if (!(fl & 1)) {
                sc = 1.0f - sqrtf(ssd / (float)a.npx) / 255.0f;
                sc = fminf(fmaxf(sc, 0.0f), 1.0f);
            }
            a.score[kp] = sc;
        }
        if (a.fb) a.fb[kp] = fb;
        unsigned long long* cnt = a.counters + 4 * (kp & (VO_COUNTER_SHARDS - 1));
        atomicAdd(&cnt[0], (unsigned long long)iters);
        atomicAdd(&cnt[1], 1ull);
        atomicAdd(&cnt[2], (unsigned long long)setups);
#ifdef VO_KLT_COUNT_CELLS // profiling builds only: slot 3 belongs to the pose step's hypothesis count
        atomicAdd(&cnt[3], (unsigned long long)cells);
#endif
    }
}

// The consumers' side of a head (vo_dev_wait_head): one lane polls the flag until the launch's epoch is there.  Bounded:
// after ~0.25 s of polling it gives up and raises *timed_out (the launch it waits for was never enqueued, or sits behind
// this kernel in the same hardware queue) -- every wave of this kernel reaches its end whatever happens.
__global__ __launch_bounds__(64) void k_wait_head(const unsigned* flag, unsigned epoch, unsigned* timed_out, const int32_t* n_dev, int32_t* host_fault) {
    if (threadIdx.x != 0) return;
    if (n_dev && *n_dev <= 0) return;   // the launch had no keypoint at all (its count lives on the device): nobody arrives
    for (int spin = 0; spin < 250000; ++spin) {
        const unsigned f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(f - epoch) >= 0) return;
        __builtin_amdgcn_s_sleep(32);   // ~2k cycles: the poll costs the memory system nothing
    }
    __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (host_fault) __hip_atomic_store(host_fault, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word of the frame loop
}

template <class C, bool SCORE>
void launch_s(const LKArgs& a, hipStream_t st) {
    // the window that cuts only the last column / row of blocks the way WinMask<true> assumes
    if (a.wu == (C::LU - 1) * C::BW + 1 && a.wv == C::LV * C::BH - 1) hipLaunchKernelGGL((k_lk<C, true, SCORE>), dim3(a.n), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_lk<C, false, SCORE>), dim3(a.n), dim3(64), 0, st, a);
}
template <class C>
void launch(const LKArgs& a, hipStream_t st) {
    if (a.score) launch_s<C, true>(a, st);
    else launch_s<C, false>(a, st);
}

} // namespace

int vo_launch_wait_head(vo_ctx* ctx, const int32_t* d_n, int32_t* host_fault) {
    unsigned* w = reinterpret_cast<unsigned*>(vo_int_head_words(ctx));
    hipLaunchKernelGGL(k_wait_head, dim3(1), dim3(64), 0, ctx->stream, w + 1, ctx->head_epoch, w + 2, d_n, host_fault);
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}

int vo_launch_klt(vo_ctx* ctx, const vo_pyr* A, const vo_pyr* B, const double* d_pu, const double* d_pv, int n,
                  int wu, int wv, int max_iters, double eps, double max_bidir, double* d_qu, double* d_qv,
                  uint8_t* d_valid, float* d_score, float* d_fb, const int32_t* d_n, int n_head) {
    if (n <= 0) return VO_OK;
    if (!(wu & 1) || !(wv & 1) || wu < 3 || wv < 3 || max_iters < 1) { ctx->last_error = "klt: BlockSize must be odd >= 3, MaxIterations >= 1"; return VO_ERR_BAD_ARG; }
    if (A->nu != B->nu || A->nv != B->nv || A->levels != B->levels) { ctx->last_error = "klt: frame shapes differ"; return VO_ERR_BAD_ARG; }
    LKArgs a;
    a.A = A->I; a.B = B->I;
    a.nu0 = A->nu; a.nv0 = A->nv;
    a.levels = A->levels; a.wu = wu; a.wv = wv;
    a.npx = wu * wv; a.max_iters = max_iters; a.n = n;
    a.eps2 = (float)(eps * eps); a.max_bidir = max_bidir;
    a.flat_thr = 64.0 * 2.0 * (double)a.npx * (double)VO_MIN_EIG;
    a.pu = d_pu; a.pv = d_pv; a.qu = d_qu; a.qv = d_qv; a.valid = d_valid; a.score = d_score; a.fb = d_fb;
    a.counters = ctx->d_counters; a.n_dev = d_n;
    a.n_head = 0; a.head_epoch = 0; a.head_cnt = a.head_flag = nullptr;
    if (n_head > 0) {
        unsigned* w = reinterpret_cast<unsigned*>(vo_int_head_words(ctx));
        a.n_head = n_head < n ? n_head : n; a.head_epoch = ++ctx->head_epoch; a.head_cnt = w; a.head_flag = w + 1;
    }
    if (wu <= CfgS::MAXU && wv <= CfgS::MAXV) launch<CfgS>(a, ctx->stream);
    else if (wu <= CfgM::MAXU && wv <= CfgM::MAXV) launch<CfgM>(a, ctx->stream);
    else if (wu <= CfgL::MAXU && wv <= CfgL::MAXV) launch<CfgL>(a, ctx->stream);
    else if (wu <= CfgX::MAXU && wv <= CfgX::MAXV) launch<CfgX>(a, ctx->stream);
    else { ctx->last_error = "klt: BlockSize larger than 55 x 55 is not supported"; return VO_ERR_UNSUPPORTED; }
    VO_HIP_CHECK(ctx, hipGetLastError());
    return VO_OK;
}
