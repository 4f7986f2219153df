// f16x3 implicit GEMM on PLANE-FORMAT operands (round 3): the arithmetic of gemm_split.hip's F16 form without its per-tile conversion.
//
//   An fp32 tensor X lives in memory as two fp16 planes  X = h + l / 2048,  h = RNE16(X), l = RNE16(2048 (X - h))  (22 significand
//   bits, see gemm_split.hip).  Weights are split ONCE at load; activations are written in plane format by the kernel that produces
//   them (GEMM / LayerNorm / attention epilogues, cbx_split_planes_f32) -- the same 4 bytes per element as fp32.  The consumer's K loop
//   is then a plain fp16 MFMA loop:  acc += Ah Bh,  accc += Ah Bl + Al Bh,  C = acc + accc / 2048 (+ epilogue), with
//     * operand tiles moved global -> LDS by global_load_lds (16 B per lane, no VGPR round trip, no VALU, no ds_write), NS = 2 or 3 LDS
//       stages, ONE barrier per K tile: the loads of tiles t+1 (.. t+NS-1) are in flight while tile t is multiplied; the wait in front of
//       the barrier is a COUNTED vmcnt (raw s_barrier: __syncthreads() would drain every DMA in flight);
//     * the LDS image lane-linear as the DMA writes it, bank conflicts removed by an XOR swizzle applied to the SOURCE address and to
//       the ds_read_b128 address (cdna_hip_programming.md rule 21): chunk c of row r sits at slot r*CH + (c ^ f(r)),
//       f(r) = (r / (16 / CH)) & (CH - 1), CH = 16-byte chunks per row and plane (BK / 8);
//     * masked rows (causal-conv left padding, rows past a ragged length) come back as zeros from the buffer descriptor's bounds check
//       (buffer_load ... lds): every lane of every load instruction writes its LDS slot, no address arithmetic or predication in the loop.
//   Same implicit-GEMM address map as gemm_split.hip for Linear and Conv1d (taps, dilation, stride, left pad, per-batch lengths);
//   no upsampling, W in [N][K] layout, K % BK == 0, Cin % BK == 0.
//
// Replaces F.linear / F.conv1d of the CFM estimator (reference models/s3gen/decoder.py:243-333, matcha/transformer.py:243-316,
// matcha/decoder.py:56-61).
#include <stdlib.h>
#include "cbx_common.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) unsigned char* lds_u8_t;  // LDS addresses stay in their address space: a generic -> LDS cast per DMA costs a null check (s_cmp + s_cselect)

// waves per SIMD the register allocation must leave room for: as many workgroups per CU as the LDS admits (two co-resident workgroups
// overlap one's epilogue with the other's K loop), at most 4 waves per SIMD (128 VGPRs)
template <int BM, int BN, int BK, int NS, int NWV, int LD>
struct PlOcc {
    static constexpr int occ = 160 * 1024 / (NS * 2 * (BM + BN) * (BK / 8) * 16);
    static constexpr int w = LD ? (NWV + LD + 3) / 4 : occ * NWV / 4;  // loader-wave form: one workgroup (NWV + LD waves) per CU
    static constexpr int waves_per_simd = w > 4 ? 4 : w < 1 ? 1 : w;
};

// One 1 KiB global -> LDS DMA (16 B per lane, LDS destination lane-linear from `lds`).  A separate __device__ function on purpose: with the
// builtin called from a lambda of the kernel template, hipcc (ROCm 7.2) silently drops the kernel's HOST stub (undefined symbol at dlopen).
__device__ __forceinline__ void pl_dma16(const __amdgpu_buffer_rsrc_t rs, lds_u8_t lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, 0, 0, 0);
}

// One loader wave of the LD > 0 forms: issues DMAs [LB, LB + LPW) of every K tile of this workgroup's tiles (LB a compile-time constant per loader
// wave: which operand / plane / rows a load moves, its LDS destination and its descriptor are then all immediates or per-lane constants -- with a
// run-time first load the issue loop of the ONE loader wave of the round-3 form took twice as long per K tile, and that wave is the form's critical
// path: profiles/r05_bench_planes_loader_waves_first.log, tile 21), waits for them with counted vmcnt, and meets the consumers at the barrier.
// The address map is the one of the symmetric form below (same slots, same swizzle).
template <int BM, int BN, int BK, int NS, int LPW, int LB>
__device__ __forceinline__ void pl_loader_wave(const cbx_gemm_pl_t& p, const lds_u8_t smem, const int lane) {
    constexpr int CH = BK / 8, RPS = 16 / CH, PLANE_SLOTS = (BM + BN) * CH, STAGE_BYTES = 2 * PLANE_SLOTS * 16;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int tiles_z = ntn * ntm, total = tiles_z * p.nz1;
    const int nk = p.K / BK;
    int vbase[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int s = (LB + i) * 64 + lane;
        const int q = s / PLANE_SLOTS, rs = s % PLANE_SLOTS;
        const int row = rs / CH, pc = rs % CH;
        const int c = pc ^ ((row / RPS) & (CH - 1));
        vbase[i] = row < BM ? (row * p.stride * (int)p.lda + (q ? (int)p.a_lo : 0) + c * 8) * 2
                            : ((row - BM) * (int)p.ldw + (q ? (int)p.w_lo : 0) + c * 8) * 2;
    }
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    int offA = 0, offW = 0;
    const int a_wstep2 = (int)(((long)p.dil * p.lda - p.Cin) * 2);
    auto load_desc = [&](int vt) {
        const int z = vt / tiles_z;
        const int t = cbx_xcd_remap(vt - z * tiles_z, tiles_z);
        const int n0 = (t % ntn) * BN, m0 = (t / ntn) * BM;
        const int lim = p.lens ? min(p.Tin, p.lens[z]) : p.Tin;
        a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(reinterpret_cast<const _Float16*>(p.A) + (long)z * p.a_s1), 0,
                                                 lim > 0 ? (int)((long)lim * p.lda * 2) : 0, 0x00020000);
        w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(reinterpret_cast<const _Float16*>(p.W) + (long)z * p.w_s1), 0,
                                                 (int)((long)p.N * p.ldw * 2), 0x00020000);
        offA = (m0 * p.stride - p.pad_left) * (int)p.lda * 2;
        offW = n0 * (int)p.ldw * 2;
    };
    int l_tile = blockIdx.x, l_kt = 0, l_g = 0, ld_c0 = 0, c_g = 0;
    if (l_tile < total) load_desc(l_tile);
#ifdef CBX_DIAG  // scripts/diag_planes.sh: 1 = no DMA after the first NS - 1 K tiles, 8 = the loader waves at s_setprio 3
    const int ldg = p.reserved0;
    if (ldg & 8) __builtin_amdgcn_s_setprio(3);
#else
    constexpr int ldg = 0;
#endif
    auto issue = [&]() {
        const lds_u8_t dst = smem + (l_g % NS) * STAGE_BYTES + LB * 1024;
        if (!(ldg & 1) || l_g < NS - 1) {
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const bool isA = ((LB + i) * 64) % PLANE_SLOTS < BM * CH;  // folds: LB, i are constants
            if (isA) pl_dma16(a_rs, dst + i * 1024, vbase[i] + offA);
            else pl_dma16(w_rs, dst + i * 1024, vbase[i] + offW);
        }
        }
        ++l_g;
        if (++l_kt == nk) {
            l_kt = 0;
            ld_c0 = 0;
            l_tile += gridDim.x;
            if (l_tile < total) load_desc(l_tile);
            return;
        }
        ld_c0 += BK;
        const bool wrap = ld_c0 >= p.Cin;
        ld_c0 = wrap ? 0 : ld_c0;
        offA += BK * 2 + (wrap ? a_wstep2 : 0);
        offW += BK * 2;
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (l_tile < total) issue();
    for (int c_tile = blockIdx.x; c_tile < total; c_tile += gridDim.x)
        for (int kt = 0; kt < nk; ++kt) {
            const int younger = l_g - c_g - 1;  // K tiles issued after the one released now: they stay in flight
            if (NS >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW > 63 ? 63 : 2 * LPW) : "memory");
            else if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW > 63 ? 63 : LPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (l_tile < total) issue();
            ++c_g;
        }
}

// ---- LayerNorm of the finished row in the epilogue (round 5, cbx_gemm_pl_t.ln_w): the ROW-SPANNING tile 64 x 256 (N == 256: attention out-projection, ff2, 1x1
// residual conv of the CFM estimator) finishes whole rows, so it can produce nn.LayerNorm(row) -- the input of the NEXT Linear -- itself: mean and variance of the
// 256 finished values of a row (the 2 x 16 values a lane holds are reduced over its 16-lane DPP row, then over the 8 row pieces of the 4 N-waves through LDS), the two-pass
// form of norm.hip (mean first, then the variance of the centred values), (v - mean) * rstd * w[n] + b[n], written in plane format by the same lane-pair exchange as the
// plain plane output.  Replaces a cbx_layernorm_planes_f32 launch (16 MB read + 16 MB written per 16000 x 256) per Linear.
constexpr int PL_ACT_LNF = 100;  // internal value of the ACT template parameter: "no activation, LayerNorm of the row to LNP"

__device__ __forceinline__ float pl_row16_sum(float x) {  // all-reduce over the 16 lanes of a DPP row: quad xor 1, xor 2, then rotations by 4 and 8
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, true));  // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));  // row_ror:8
    return x;
}

// ---- DEFERRED epilogue (round 6, DF = true; loader-wave forms only).  The K = 256 Linears of the CFM transformer blocks (q | k | v: N = 1536, ff1: N = 1024) run
// four K tiles of MFMAs per output tile and then an epilogue of the same length (fp32 -> planes: ~9 VALU per element, GELU: ~35, and 4-byte stores) during which the
// matrix pipe of the SIMD idles, because the 8 consumer waves of the workgroup reach their epilogues together.  In this form a finished tile is only FOLDED
// (v = acc + accc / 2048 + bias: 32 registers per wave instead of 64) and the rest of its epilogue -- activation, residual, plane split, lane exchange, stores -- is cut
// into QUADS (the four accumulator registers 4q .. 4q + 3 of one 32 x 32 sub-tile: four consecutive rows per lane) that are issued BETWEEN the MFMA groups of the next
// tile's first four K tiles: VALU and stores of a wave issue while the matrix pipe works on its partner's (and its own) MFMAs (MI355X_MICROARCH.md, "Two waves per SIMD":
// the pipe is paced at 32 cycles per MFMA, VALU slots in between are free).  Same operations on the same values in the same order per element: bit-identical output.
// DF = 1: tiles that write planes (P, and PT for the transposed column range) and nothing else -- q | k | v, ff1; DF = 2: tiles that write fp32 C (+ residual R) and nothing
// else -- out-projection, ff2, convs.  The kind is a template parameter so that a quad is straight-line code and needs two buffer descriptors, not four.
struct PlEpi {
    __amdgpu_buffer_rsrc_t a_rs, b_rs;  // DF = 1: P, PT;  DF = 2: C, R
    int m0, n0;
    int zq, t0;                         // DF = 1, transposed tiles: m0 / pt_T and m0 % pt_T
    bool vt, hasR;
};

template <int WM, int WN>
__device__ __forceinline__ void pl_quad_res(const cbx_gemm_pl_t& p, const PlEpi& e, int wm, int wn, int lr, int lh, int i, int j, int q, float (&res)[4]) {
    const int n = e.n0 + wn * WN + j * 32 + lr;
    const int mb = e.m0 + wm * WM + i * 32 + 4 * lh;
    const int ldr4 = (int)p.ldr * 4;
    const int ro = n < p.N ? mb * ldr4 + n * 4 : (int)0x80000000;
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) res[ee] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(e.b_rs, ro, (8 * q + ee) * ldr4, 0));
}

template <int ACT, int KIND, int BM, int WM, int WN>
__device__ __forceinline__ void pl_quad_epi(const cbx_gemm_pl_t& p, const PlEpi& e, int wm, int wn, int lr, int lh, int i, int j, int q, const float (&vin)[4],
                                            const float (&res)[4], float& amax) {
    constexpr int OOB = (int)0x80000000;
    const int n = e.n0 + wn * WN + j * 32 + lr;
    const bool nok = n < p.N;
    const int ml = wm * WM + i * 32 + 4 * lh;  // row of register 0 of the sub-tile inside the tile
    const int mb = e.m0 + ml;
    float v[4];
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) v[ee] = vin[ee];
    if constexpr (ACT == CBX_ACT_GELU_ERF) {
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) v[ee] = cbx_gelu_erf(v[ee]);
    } else if constexpr (ACT == CBX_ACT_SILU) {
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) v[ee] = v[ee] / (1.0f + __expf(-v[ee]));
    }
    if constexpr (KIND == 2) {
        if (e.hasR) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) v[ee] += res[ee];
        }
        const int ldc4 = (int)p.ldc * 4;
        const int co = nok ? mb * ldc4 + n * 4 : OOB;
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[ee]), e.a_rs, co, (8 * q + ee) * ldc4, 0);
    } else if (e.vt) {  // four consecutive tokens of the lane's column: one 8-byte store per plane (see the plain epilogue)
        unsigned h01, l01, h23, l23;
        cbx_split2(v[0], v[1], h01, l01);
        cbx_split2(v[2], v[3], h23, l23);
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[0]), "v"(v[1]));
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[2]), "v"(v[3]));
        const int m = mb + 8 * q;
        int zz, t;
        if (p.pt_T >= BM) {  // a tile crosses at most one group boundary: m / pt_T from the tile's quotient (same values as the division)
            const int tl = e.t0 + ml + 8 * q;
            const bool over = tl >= p.pt_T;
            zz = e.zq + (over ? 1 : 0);
            t = tl - (over ? p.pt_T : 0);
        } else {
            zz = m / p.pt_T;
            t = m - zz * p.pt_T;
        }
        const long eo = (long)zz * p.pt_zs + (long)(n - p.pt_n0) * p.pt_ld + t;
        const int bo = (nok && m < p.M) ? (int)(eo * 2) : OOB;
        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u32x2v{h01, h23}, e.b_rs, bo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2v{l01, l23}, e.b_rs, bo + (int)p.pt_lo * 2, 0, 0);
    } else {  // the lane-pair exchange of the plain epilogue
        const int ldp2 = (int)p.ldp * 2;
        const bool odd = lr & 1;
        const int ne = n & ~1;
        const int po = (ne < p.N ? mb * ldp2 + ne * 2 : OOB) + (odd ? ldp2 : 0);
        const int plo2 = (int)p.p_lo * 2;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const float give = odd ? v[r] : v[r + 1];
            const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
            const float c0 = odd ? got : v[r], c1 = odd ? v[r + 1] : got;
            unsigned h2, l2;
            cbx_split2(c0, c1, h2, l2);
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(c0), "v"(c1));
            const int so = (8 * q + r) * ldp2;
            __builtin_amdgcn_raw_buffer_store_b32(h2, e.a_rs, po, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(l2, e.a_rs, po + plo2, so, 0);
        }
    }
}

// LD > 0 = loader-wave form: the workgroup has LD EXTRA waves that do nothing but issue the DMAs of every K tile and wait for them; the
// NWV consumer waves only read LDS, multiply and run the epilogue.  A vector-memory instruction costs its issuing wave 100-200 cycles while
// the CU's address path is busy, and a wave issues in order: in the symmetric form every wave's MFMAs queue behind its own DMA issue
// (measured: DMA time and MFMA time ADD, profiles/r03_planes_diag_switches.log); with dedicated loaders they overlap.  ONE loader wave issues
// a 1 KiB DMA per ~100 cycles -- MI355X_MICROARCH.md `ldsdma-fill`: ~25 GB/s per CU, 6.4 TB/s over the chip, which is the operand stream at
// ~8 TB/s that every round-3 tile form ran into -- while the MFMAs of a 128 x 128 x 64 K tile want 64 KiB per ~1500 cycles (~100 GB/s per
// CU): round 5 adds LD = 2 and LD = 4 (one loader per SIMD), each loader wave issuing 1 / LD of a K tile's DMAs (for LD = 4 exactly one of
// {A.h, W.h, A.l, W.l}).
template <int BM, int BN, int WARPS_M, int WARPS_N, int BK, int NS, int ACT, int LD, int DF = 0>
// (the parentheses keep the template commas away from the variadic __launch_bounds__ macro)
__global__ __launch_bounds__((WARPS_M * WARPS_N + LD) * 64, (PlOcc<BM, BN, BK, NS, WARPS_M * WARPS_N, LD>::waves_per_simd))
void gemm_pl_kernel(const cbx_gemm_pl_t p, int* range_flag) {
    static_assert(DF == 0 || (LD > 0 && ACT != PL_ACT_LNF), "the deferred epilogue belongs to the loader-wave forms (the consumers issue no DMA: no counted vmcnt waits to disturb)");
    constexpr int NWV = WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int CH = BK / 8;                      // 16-byte chunks per row and plane
    constexpr int RPS = 16 / CH;                    // rows per swizzle step
    constexpr int PLANE_SLOTS = (BM + BN) * CH;     // A rows then B rows
    constexpr int STAGE_SLOTS = 2 * PLANE_SLOTS;    // plane h, plane l
    constexpr int STAGE_BYTES = STAGE_SLOTS * 16;
    constexpr int NLOAD = STAGE_SLOTS / 64;         // wave-level DMA instructions (1 KiB each) per stage
    static_assert((BM * CH) % 64 == 0 && (BN * CH) % 64 == 0, "a DMA instruction must not straddle the A / B boundary");
    static_assert(LD ? NLOAD % LD == 0 : NLOAD % NWV == 0, "DMA instructions must divide evenly over the (loader) waves");
    static_assert(WM % 32 == 0 && WN % 32 == 0 && (CH == 4 || CH == 8) && LD >= 0 && LD <= 4, "tile shape");
    constexpr int LPW = LD ? NLOAD / LD : NLOAD / NWV;  // DMA instructions per K tile issued by one (loader) wave

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const lds_u8_t sm3 = (lds_u8_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WARPS_N, wn = wid % WARPS_N;
    const bool is_loader = LD && wid >= NWV;
    const int lbase = LD ? 0 : wid * LPW;  // first DMA instruction (of a stage) this wave issues (symmetric form; the loader waves: pl_loader_wave)
    // ---- persistent tile loop: workgroup b owns the virtual tiles b, b + gridDim.x, ...; the LOADER side (DMA issue) runs NS - 1 K tiles ahead
    //      of the CONSUMER side (MFMA + epilogue) straight across tile boundaries, so the first K tiles of the next output tile are in flight
    //      while this one's epilogue computes and stores, and no tile but a workgroup's first pays the load latency.  With gridDim.x = number
    //      of tiles this is the plain one-tile-per-workgroup kernel.
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int tiles_z = ntn * ntm, total = tiles_z * p.nz1;
    const int nk = p.K / BK;
    auto tile_of = [&](int vt, int& z, int& m0, int& n0) {
        z = vt / tiles_z;
        const int t = cbx_xcd_remap(vt - z * tiles_z, tiles_z);  // tiles that share an A panel run on one XCD (gridDim.x % 8 == 0 keeps vt % 8 = b % 8)
        n0 = (t % ntn) * BN;
        m0 = (t / ntn) * BM;
    };

    // ---- DMA descriptors.  Load i of this wave fills slots [L*64, L*64 + 64) of a stage, L = wid * LPW + i; the lane's slot fixes (plane, row,
    //      chunk) for the whole K walk of a tile, so a load's address is ONE 32-bit byte offset per lane: a per-lane constant (vbase) plus
    //      wave-uniform tile and K / tap terms.  Validity is the buffer descriptor's business (the bounds check looks at this vector offset;
    //      a scalar offset would bypass it): rows >= lim lie behind num_records, rows < 0 (causal left padding) wrap to offsets >= 2^31 --
    //      both come back as zeros -- and rows of the M / N tail read whatever is there (their results are never stored).
    int vbase[LPW];
    int offA = 0, offW = 0;  // wave-uniform parts of the A / W offsets: tile origin + K / tap advance (added in the VALU: see above)
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int s = (lbase + i) * 64 + lane;
        const int q = s / PLANE_SLOTS, rs = s % PLANE_SLOTS;
        const int row = rs / CH, pc = rs % CH;
        const int c = pc ^ ((row / RPS) & (CH - 1));
        vbase[i] = row < BM ? (row * p.stride * (int)p.lda + (q ? (int)p.a_lo : 0) + c * 8) * 2
                            : ((row - BM) * (int)p.ldw + (q ? (int)p.w_lo : 0) + c * 8) * 2;
    }
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    const int a_wstep2 = (int)(((long)p.dil * p.lda - p.Cin) * 2);  // extra byte step of the A offset at a tap wrap
    auto load_desc = [&](int vt) {
        int z, m0, n0;
        tile_of(vt, z, m0, n0);
        const int lim = p.lens ? min(p.Tin, p.lens[z]) : p.Tin;
        a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(reinterpret_cast<const _Float16*>(p.A) + (long)z * p.a_s1), 0,
                                                 lim > 0 ? (int)((long)lim * p.lda * 2) : 0, 0x00020000);
        w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(reinterpret_cast<const _Float16*>(p.W) + (long)z * p.w_s1), 0,
                                                 (int)((long)p.N * p.ldw * 2), 0x00020000);
        offA = (m0 * p.stride - p.pad_left) * (int)p.lda * 2;
        offW = n0 * (int)p.ldw * 2;
    };
    int l_tile = blockIdx.x, l_kt = 0, l_g = 0;  // loader position: tile, K tile inside it, K tiles issued so far (ring position)
    int ld_c0 = 0;                               // channel block inside the current conv tap
    if (!LD && l_tile < total) load_desc(l_tile);  // (the loader waves of the LD forms keep their own descriptors: pl_loader_wave)
    auto issue = [&]() {  // the loader's next K tile into ring stage l_g % NS
        const lds_u8_t dst = sm3 + (l_g % NS) * STAGE_BYTES + lbase * 1024;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const bool isA = ((lbase + i) * 64) % PLANE_SLOTS < BM * CH;
            if (isA) pl_dma16(a_rs, dst + i * 1024, vbase[i] + offA);
            else pl_dma16(w_rs, dst + i * 1024, vbase[i] + offW);
        }
        ++l_g;
        if (++l_kt == nk) {  // next output tile of this workgroup
            l_kt = 0;
            ld_c0 = 0;
            l_tile += gridDim.x;
            if (l_tile < total) load_desc(l_tile);
            return;
        }
        ld_c0 += BK;
        const bool wrap = ld_c0 >= p.Cin;  // next K tile starts the next conv tap
        ld_c0 = wrap ? 0 : ld_c0;
        offA += BK * 2 + (wrap ? a_wstep2 : 0);
        offW += BK * 2;
    };

    f32x16 acc[TM][TN], accc[TM][TN];
    const int lr = lane & 31, lh = lane >> 5;
    const int swz = (lr / RPS) & (CH - 1);  // rows of a fragment are base + lr with base % 32 == 0: f(row) = f(lr)
    const int a_off = (wm * WM + lr) * CH * 16, b_off = (BM + wn * WN + lr) * CH * 16;
    // (round 6, measured and not kept: a second operand register set with the requests of K chunk kc + 1 pinned in front of the MFMA group of chunk kc -- 138 instead of
    // 111 VGPRs, within 1 - 3 % on every shape: the K loop's MFMA phase already runs at ~80 % of the clock-adjusted matrix rate, profiles/r06_r_plane_gemm_operand_prefetch.log)
    auto compute_h = [&](int stage, auto&& after_kc) {  // after_kc(kc): instructions placed behind the MFMA group of K chunk kc (the deferred epilogue's quads)
        const unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
            const int co = ((kc * 2 + lh) ^ swz) << 4;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(st + a_off + i * 32 * CH * 16 + co);
                al[i] = *reinterpret_cast<const f16x8*>(st + PLANE_SLOTS * 16 + a_off + i * 32 * CH * 16 + co);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(st + b_off + j * 32 * CH * 16 + co);
                bl[j] = *reinterpret_cast<const f16x8*>(st + PLANE_SLOTS * 16 + b_off + j * 32 * CH * 16 + co);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accc[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            after_kc(kc);
        }
    };
    auto compute = [&](int stage) { compute_h(stage, [](int) {}); };

    // ---- main loop: NS LDS stages, one barrier per K tile.  Before the barrier every wave waits until ITS DMAs of the K tile about to be
    //      consumed have landed -- a COUNTED wait: vector-memory operations complete in issue order, so "at most n outstanding" with n = the
    //      number of operations issued after those DMAs (younger K tiles, and the previous tile's epilogue loads / stores) is exact and leaves
    //      everything younger in flight; after the barrier the K tile is complete for everybody and nobody reads the stage the next issue
    //      overwrites (it held the K tile consumed one iteration ago).
    static_assert(NS >= 2 && NS <= 4 && (LD || NS <= 3), "two or three LDS stages (the loader-wave form: up to four)");
#ifdef CBX_DIAG  // scripts/diag_planes.sh: parts of the kernel switched off (1 no DMA after the prologue, 2 no ds_read / MFMA, 4 no epilogue stores)
    const int dg = p.reserved0;
#define DG(bit) (dg & (bit))
#else
#define DG(bit) 0
#endif
    constexpr int EPI_OPS = TM * TN * 16;  // vector-memory operations every epilogue issues at least (one output kind: C or P)
    [[maybe_unused]] constexpr int W_EPI0 = (NS - 2) * LPW + EPI_OPS > 63 ? 63 : (NS - 2) * LPW + EPI_OPS;  // first K tile after an epilogue
    [[maybe_unused]] constexpr int W_EPI1 = LPW + EPI_OPS > 63 ? 63 : LPW + EPI_OPS;                          // NS = 3: second K tile after an epilogue
    // ... except the epilogue of a TRANSPOSED tile (PT, ABI v8), which issues 8-byte stores: HALF as many operations.  Counting 16 per
    // 32 x 32 tile behind it would let up to TM * TN * 8 of the OLDER operations -- the DMAs of the K tile about to be consumed -- stay
    // outstanding.  (Found by the emulator's deferred-DMA mode, tests/simt/: never observed on the GPU, where an epilogue outlasts a DMA.)
    constexpr int EPI_OPS_T = TM * TN * 8;
    [[maybe_unused]] constexpr int W_EPI0T = (NS - 2) * LPW + EPI_OPS_T > 63 ? 63 : (NS - 2) * LPW + EPI_OPS_T;
    [[maybe_unused]] constexpr int W_EPI1T = LPW + EPI_OPS_T > 63 ? 63 : LPW + EPI_OPS_T;
    bool prev_t = false;  // the previous tile of this workgroup was a transposed one
    int c_g = 0;  // consumer ring position
    if (is_loader) {  // ---- a loader wave: wait for ITS share of K tile g, release it to the consumers at the barrier, issue its share of K tile g + NS - 1
        if constexpr (LD > 0) {
            constexpr int LL = NLOAD / (LD > 0 ? LD : 1);
            const int li = wid - NWV;
            if (li == 0) pl_loader_wave<BM, BN, BK, NS, LL, 0>(p, sm3, lane);
            if constexpr (LD >= 2) { if (li == 1) pl_loader_wave<BM, BN, BK, NS, LL, LL>(p, sm3, lane); }
            if constexpr (LD >= 3) { if (li == 2) pl_loader_wave<BM, BN, BK, NS, LL, 2 * LL>(p, sm3, lane); }
            if constexpr (LD >= 4) { if (li == 3) pl_loader_wave<BM, BN, BK, NS, LL, 3 * LL>(p, sm3, lane); }
        }
        return;
    }
    if (!LD) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (l_tile < total) issue();
    }
    // ---- deferred epilogue (DF): the folded values of the previous tile and where they go; quad qd = (sub-tile i, j; registers 4q .. 4q + 3)
    constexpr int NQ = TM * TN * 4;          // quads per wave and tile
    constexpr int KCN = BK / 16, PKT = 4;    // K chunks per K tile; the quads are spread over the first PKT K tiles of the next tile
    constexpr int SLOTS = PKT * KCN;
    [[maybe_unused]] float pend[TM][TN][16];
    [[maybe_unused]] float pres[4] = {0.f, 0.f, 0.f, 0.f};  // residual values of the NEXT quad (requested one quad ahead)
    [[maybe_unused]] PlEpi pe;
    [[maybe_unused]] bool has_pend = false;
    [[maybe_unused]] float amax_df = 0.f;
    [[maybe_unused]] auto quad = [&](int qd) {
        const int i = qd / (TN * 4), j = (qd / 4) % TN, q = qd & 3;
        const float vin[4] = {pend[i][j][4 * q], pend[i][j][4 * q + 1], pend[i][j][4 * q + 2], pend[i][j][4 * q + 3]};
        pl_quad_epi<ACT, DF, BM, WM, WN>(p, pe, wm, wn, lr, lh, i, j, q, vin, pres, amax_df);
        if (DF == 2 && pe.hasR && qd + 1 < NQ) pl_quad_res<WM, WN>(p, pe, wm, wn, lr, lh, (qd + 1) / (TN * 4), ((qd + 1) / 4) % TN, (qd + 1) & 3, pres);
    };
    [[maybe_unused]] auto slot_quads = [&](int s) {  // the quads of slot s of SLOTS: an even spread, in order
#pragma unroll
        for (int qd = (s * NQ + SLOTS - 1) / SLOTS; qd < ((s + 1) * NQ + SLOTS - 1) / SLOTS; ++qd) quad(qd);
    };
    for (int c_tile = blockIdx.x; c_tile < total; c_tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = accc[i][j][r] = 0.f;
        const bool after_epi = c_tile != (int)blockIdx.x;
        int kt = 0;
        if constexpr (DF) {
            if (has_pend) {  // (wave-uniform) the previous tile's quads ride on this tile's first PKT K tiles
#pragma unroll
                for (int u = 0; u < PKT; ++u) {
                    if (kt < nk) {
                        __builtin_amdgcn_s_barrier();
                        compute_h(c_g % NS, [&](int kc) { slot_quads(u * KCN + kc); });
                        ++c_g;
                        ++kt;
                    } else {
#pragma unroll
                        for (int kc = 0; kc < KCN; ++kc) slot_quads(u * KCN + kc);
                    }
                }
                has_pend = false;
            }
        }
        for (; kt < nk; ++kt) {
            if (!LD) {
                const int younger = l_g - c_g - 1;  // K tiles issued after the one consumed now (0 .. NS - 2)
                if (after_epi && kt == 0 && younger == NS - 2 && !DG(5)) {
                    if (prev_t) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_EPI0T) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_EPI0) : "memory");
                } else if (NS == 3 && after_epi && kt == 1 && younger == 1 && !DG(5)) {
                    if (prev_t) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_EPI1T) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_EPI1) : "memory");
                }
                else if (NS == 3 && younger == 1 && !DG(1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (!LD && l_tile < total && !DG(1)) issue();
            if (!DG(2)) compute(c_g % NS);
            ++c_g;
        }
        int z, m0, n0;
        tile_of(c_tile, z, m0, n0);
        if (DG(16)) continue;  // (diagnostic build) no epilogue at all

    // ---- epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    //      Lean by construction: every store / residual load is ONE buffer instruction (32-bit lane offset + a scalar row offset); rows >= M fall
    //      behind num_records and are dropped by the hardware, lanes whose column is >= N carry an out-of-range offset; the activation
    //      and the output kinds are branched on once per tile, never per element.
    constexpr int OOB = (int)0x80000000;
    const int Mrem = p.M;
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
        p.C ? p.C + (long)z * p.c_s1 : nullptr, 0, p.C ? (int)((((long)Mrem - 1) * p.ldc + p.N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.R ? p.R + (long)z * p.r_s1 : nullptr), 0, p.R ? (int)((((long)Mrem - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
    // plane output: the l plane lies p_lo halves after the h plane of the same row, so one descriptor covers both
    const __amdgpu_buffer_rsrc_t p_rs = __builtin_amdgcn_make_buffer_rsrc(
        p.P ? reinterpret_cast<_Float16*>(p.P) + (long)z * p.p_s1 : nullptr, 0, p.P ? (int)((((long)Mrem - 1) * p.ldp + p.p_lo + (p.PT ? p.pt_n0 : p.N)) * 2) : 0, 0x00020000);
    // transposed plane output (V^T of the fused q | k | v projection): whole tiles, never a part of one (pt_n0 % 256 == 0)
    const bool vt_tile = p.PT != nullptr && n0 >= p.pt_n0;
    const __amdgpu_buffer_rsrc_t t_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<_Float16*>(p.PT), 0, p.PT ? (int)((((long)(p.M / (p.pt_T > 0 ? p.pt_T : 1)) - 1) * p.pt_zs + ((long)(p.N - p.pt_n0) - 1) * p.pt_ld + p.pt_lo + p.pt_T) * 2) : 0, 0x00020000);
    const bool hasC = p.C != nullptr && !vt_tile, hasP = p.P != nullptr && !vt_tile, hasR = p.R != nullptr && !vt_tile;
    prev_t = vt_tile;
    const int ldc4 = (int)p.ldc * 4, ldr4 = (int)p.ldr * 4, ldp2 = (int)p.ldp * 2;
    const bool odd = lr & 1;
    float amax = 0.f;
    if constexpr (ACT == PL_ACT_LNF) {
        static_assert(ACT != PL_ACT_LNF || (TM == 1 && BN == 256 && LD == 0 && WARPS_N * 2 == 8), "LayerNorm epilogue: 64 x 256 tile, 2 x 4 waves of 32 x 64");
        float* const red = reinterpret_cast<float*>(smem + NS * STAGE_BYTES);  // [WARPS_M][8 row pieces][32 rows]
        float* const stat = red + WARPS_M * 8 * 32;                              // [BM]: mean, then rstd
        const __amdgpu_buffer_rsrc_t n_rs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<_Float16*>(p.LNP) + (long)z * p.lnp_s1, 0, (int)((((long)Mrem - 1) * p.ld_lnp + p.lnp_lo + p.N) * 2), 0x00020000);
        const int mb = m0 + wm * WM + 4 * lh;  // row of register 0
        // 1. the finished values (bias, residual), kept in the accumulator registers; fp32 output
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn * WN + j * 32 + lr;  // (n0 == 0, N == 256: every column exists)
            const float bia = p.bias ? p.bias[n] : 0.f;
            const int ro = mb * ldr4 + n * 4, co = mb * ldc4 + n * 4;
            float res[16];
            if (hasR) {
#pragma unroll
                for (int r = 0; r < 16; ++r) res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, ro, ((r & 3) + 8 * (r >> 2)) * ldr4, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = __builtin_fmaf(accc[0][j][r], 1.0f / CBX_F16_LO_SCALE, acc[0][j][r]) + bia;
                if (hasR) v += res[r];
                acc[0][j][r] = v;
                if (hasC) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), c_rs, co, ((r & 3) + 8 * (r >> 2)) * ldc4, 0);
            }
        }
        // 2. mean, 3. variance of the centred values: lane -> 16-lane row -> the row's 8 pieces (4 N-waves x 2 DPP rows) through LDS
        const int piece = (wm * 8 + wn * 2 + (lr >> 4)) * 32;
        float st[16];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (pass) acc[0][j][r] -= st[r];  // centre on the mean of pass 0
                    t += pass ? acc[0][j][r] * acc[0][j][r] : acc[0][j][r];
                }
                t = pl_row16_sum(t);
                if ((lr & 15) == 0) red[piece + (r & 3) + 8 * (r >> 2) + 4 * lh] = t;
            }
            __syncthreads();
            if (tid < BM) {
                const float* rp = red + ((tid >> 5) * 8) * 32 + (tid & 31);
                float t = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) t += rp[q * 32];
                stat[tid] = pass ? rsqrtf(t * (1.0f / 256.0f) + p.ln_eps) : t * (1.0f / 256.0f);
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(stat + wm * WM + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) st[4 * g + e] = t4[e];
            }
        }
        // 4. (v - mean) * rstd * w[n] + b[n] -> planes (the lane-pair exchange of the plain plane output)
        const int ldn2 = (int)p.ld_lnp * 2, nlo2 = (int)p.lnp_lo * 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn * WN + j * 32 + lr, ne = n & ~1;
            const float gw = p.ln_w[n], gb = p.ln_b ? p.ln_b[n] : 0.f;
            const int po = mb * ldn2 + ne * 2 + (odd ? ldn2 : 0);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float y0 = __builtin_fmaf(acc[0][j][r] * st[r], gw, gb), y1 = __builtin_fmaf(acc[0][j][r + 1] * st[r + 1], gw, gb);
                const float give = odd ? y0 : y1;
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                const float c0 = odd ? got : y0, c1 = odd ? y1 : got;  // columns ne, ne + 1 of this lane's row
                unsigned h2, l2;
                cbx_split2(c0, c1, h2, l2);
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(c0), "v"(c1));
                const int so = ((r & 3) + 8 * (r >> 2)) * ldn2;
                __builtin_amdgcn_raw_buffer_store_b32(h2, n_rs, po, so, 0);
                __builtin_amdgcn_raw_buffer_store_b32(l2, n_rs, po + nlo2, so, 0);
            }
        }
        if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
    } else if constexpr (DF) {  // fold: the finished values (bias included) wait in `pend` for the next tile's K loop
        pe.a_rs = DF == 1 ? p_rs : c_rs;
        pe.b_rs = DF == 1 ? t_rs : r_rs;
        pe.m0 = m0; pe.n0 = n0;
        pe.vt = vt_tile; pe.hasR = hasR;
        if (DF == 1 && vt_tile) {
            pe.zq = m0 / p.pt_T;
            pe.t0 = m0 - pe.zq * p.pt_T;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + lr;
            const float bia = p.bias ? p.bias[n < p.N ? n : 0] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) pend[i][j][r] = __builtin_fmaf(accc[i][j][r], 1.0f / CBX_F16_LO_SCALE, acc[i][j][r]) + bia;
        }
        if (DF == 2 && hasR) pl_quad_res<WM, WN>(p, pe, wm, wn, lr, lh, 0, 0, 0, pres);
        has_pend = true;
    } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        const bool nok = n < p.N;
        const float bia = p.bias ? p.bias[nok ? n : 0] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * lh;  // row of register 0
            const int ro = nok ? mb * ldr4 + n * 4 : OOB, co = nok ? mb * ldc4 + n * 4 : OOB;
            const int ne = n & ~1;  // plane output: first column of the lane pair
            const int po = (ne < p.N ? mb * ldp2 + ne * 2 : OOB) + (odd ? ldp2 : 0);
            const int plo2 = (int)p.p_lo * 2;
            // eight accumulator registers at a time (rows 8 h + {0..3} + 4 lh, h = 0..3 -> two passes of two row quads): keeps the live set
            // inside the 128-VGPR budget that lets two workgroups share a CU
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float v[8], res[8];
                if (hasR) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, ro, (((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * ldr4, 0));
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(accc[i][j][r0 + r], 1.0f / CBX_F16_LO_SCALE, acc[i][j][r0 + r]) + bia;
                if constexpr (ACT == CBX_ACT_GELU_ERF) {  // the activation is a template parameter: one epilogue body per kernel (instruction cache)
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = cbx_gelu_erf(v[r]);
                } else if constexpr (ACT == CBX_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = v[r] / (1.0f + __expf(-v[r]));
                }
                if (hasR) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += res[r];
                }
                if (hasC && !DG(4)) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), c_rs, co, (((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * ldc4, 0);
                }
                if (vt_tile) {
                    // registers 4q .. 4q+3 of this pass are FOUR CONSECUTIVE ROWS (tokens) of the lane's column: the transposed image wants
                    // exactly that -- 4 halves of the h plane and 4 of the l plane per 8-byte store, no lane exchange.  pt_T % 4 == 0 keeps the four
                    // tokens inside one group of pt_T rows.
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned h01, l01, h23, l23;
                        cbx_split2(v[4 * q], v[4 * q + 1], h01, l01);
                        cbx_split2(v[4 * q + 2], v[4 * q + 3], h23, l23);
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[4 * q]), "v"(v[4 * q + 1]));
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[4 * q + 2]), "v"(v[4 * q + 3]));
                        const int m = mb + 8 * (r0 / 4 + q);
                        const int zz = m / p.pt_T, t = m - zz * p.pt_T;
                        const long eo = (long)zz * p.pt_zs + (long)(n - p.pt_n0) * p.pt_ld + t;
                        const int bo = (nok && m < p.M) ? (int)(eo * 2) : OOB;
                        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2v{h01, h23}, t_rs, bo, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2v{l01, l23}, t_rs, bo + (int)p.pt_lo * 2, 0, 0);
                    }
                }
                if (hasP) {
                    // The lane pair (even, odd column) owns columns n, n + 1 of every row.  Of each register pair (rows m, m + 1) the even lane
                    // converts row m and the odd lane row m + 1: one exchange gives each lane both columns of its row, which it splits into the h
                    // word and the l word (two 4-byte stores per lane and register pair instead of a conversion per element).
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const float give = odd ? v[r] : v[r + 1];
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                        const float c0 = odd ? got : v[r], c1 = odd ? v[r + 1] : got;  // columns ne, ne + 1 of this lane's row
                        unsigned h2, l2;
                        cbx_split2(c0, c1, h2, l2);
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(c0), "v"(c1));
                        if (!DG(4)) {
                            const int so = (((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * ldp2;
                            __builtin_amdgcn_raw_buffer_store_b32(h2, p_rs, po, so, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(l2, p_rs, po + plo2, so, 0);
                        }
                    }
                }
            }
        }
    }
        if ((hasP || vt_tile) && amax > 65504.f && range_flag) atomicOr(range_flag, 1);
    }  // (plain epilogue)
    }  // tile loop
    if constexpr (DF) {
        if (has_pend) {  // the last tile of this workgroup: nothing left to hide behind
#pragma unroll
            for (int qd = 0; qd < NQ; ++qd) quad(qd);
        }
        if (amax_df > 65504.f && range_flag) atomicOr(range_flag, 1);
    }
}

// x (rows, C) fp32 -> planes.  One float4 per thread; two 8-byte stores.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ P, long rows, int C4,
                                                           long ldx, long ldp, long p_lo, int* range_flag) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C4) return;
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    const f16x4 h = __builtin_convertvector(v, f16x4);
    const f32x4 t = v * CBX_F16_LO_SCALE;
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf((float)h[e], -CBX_F16_LO_SCALE, t[e]);
    *reinterpret_cast<f16x4*>(P + r * ldp + c) = h;
    *reinterpret_cast<f16x4*>(P + r * ldp + p_lo + c) = __builtin_convertvector(d, f16x4);
    float amax = 0.f;
    cbx_amax4(amax, v);
    if (amax > 65504.f && range_flag) atomicOr(range_flag, 1);
}

int g_pl_persist = 1;  // TEST HOOK: cbx_set_planes_persist

template <int BM, int BN, int WARPS_M, int WARPS_N, int BK, int NS, int ACT, int LD, int DF = 0>
int launch_pl_act(const cbx_gemm_pl_t& p, hipStream_t st) {
    constexpr size_t lds = (size_t)NS * 2 * (BM + BN) * (BK / 8) * 16 + (ACT == PL_ACT_LNF ? (size_t)(WARPS_M * 8 * 32 + BM) * 4 : 0);  // + the LayerNorm epilogue's row pieces
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = gemm_pl_kernel<BM, BN, WARPS_M, WARPS_N, BK, NS, ACT, LD, DF>;
    constexpr int THREADS = (WARPS_M * WARPS_N + LD) * 64;
    static int resident_dev[64] = {0};  // per device ordinal (hipFuncSetAttribute and the CU count are per device): workgroups the chip holds at
    int& resident = resident_dev[cbx_device()];  // once (advisory: nothing in the kernel depends on co-residency)
    if (!resident) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return cbx_set_error((int)e, "gemm_planes: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), THREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        resident = per_cu * prop.multiProcessorCount / 8 * 8;  // a multiple of the XCD count keeps the XCD affinity of the tile order
        if (resident < 8) resident = 8;
    }
    const long total = (long)((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * p.nz1;
    const long cap = g_pl_persist > 1 ? g_pl_persist : resident;  // > 1: an explicit workgroup count (tests force multi-tile walks on small shapes)
    const unsigned grid = (unsigned)(g_pl_persist && total > cap ? cap : total);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, p, cbx_range_flag());
    return cbx_check_launch("gemm_planes");
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int BK, int NS = 2, int LD = 0, int DF = 0>
int launch_pl(const cbx_gemm_pl_t& p, hipStream_t st) {
    if (p.act == CBX_ACT_GELU_ERF) return launch_pl_act<BM, BN, WARPS_M, WARPS_N, BK, NS, CBX_ACT_GELU_ERF, LD, DF>(p, st);
    if (p.act == CBX_ACT_SILU) return launch_pl_act<BM, BN, WARPS_M, WARPS_N, BK, NS, CBX_ACT_SILU, LD, DF>(p, st);
    return launch_pl_act<BM, BN, WARPS_M, WARPS_N, BK, NS, CBX_ACT_NONE, LD, DF>(p, st);
}

}  // namespace

static int g_pl_tile = 0;  // TEST HOOK: cbx_set_planes_tile (callers pass cbx_gemm_pl_t.tile per call)
extern "C" int cbx_set_planes_tile(int t) {
    g_pl_tile = t;
    return 0;
}
// tuning knob: 1 (default) = persistent workgroups (grid = what the chip holds, each workgroup walks several tiles with its DMA running
// across tile boundaries); 0 = one tile per workgroup; n > 1 = exactly n workgroups
extern "C" int cbx_set_planes_persist(int on) {
    g_pl_persist = on;
    return 0;
}

extern "C" int cbx_gemm_planes(const cbx_gemm_pl_t* pp, void* stream) {
    cbx_gemm_pl_t p = *pp;
    if (p.taps < 1) p.taps = 1;
    if (p.dil < 1) p.dil = 1;
    if (p.stride < 1) p.stride = 1;
    if (p.nz1 < 1) p.nz1 = 1;
    if (p.Cin <= 0) p.Cin = p.K / p.taps;
    if (p.Tin <= 0) p.Tin = p.M;
    CBX_REQUIRE(p.alpha == 0.f || p.alpha == 1.f, "gemm_planes: alpha is not supported (must be 0 or 1)");
    CBX_REQUIRE(p.tile >= CBX_PL_TILE_CORESIDENT && p.tile <= 64, "gemm_planes: tile=%d (0 automatic, 1 .. the tile menu, -1 co-resident)", p.tile);
    CBX_REQUIRE(p.act == CBX_ACT_NONE || p.act == CBX_ACT_GELU_ERF || p.act == CBX_ACT_SILU, "gemm_planes: activation %d is not served (none, GELU (erf), SiLU)", p.act);
    CBX_REQUIRE(p.A && p.W && (p.C || p.P), "gemm_planes: null operand");
    CBX_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.K == p.taps * p.Cin, "gemm_planes: bad shape M=%d N=%d K=%d taps=%d Cin=%d", p.M, p.N, p.K, p.taps, p.Cin);
    CBX_REQUIRE(p.Cin % 32 == 0, "gemm_planes: Cin=%d must be a multiple of 32 (a K tile must not straddle two conv taps)", p.Cin);
    CBX_REQUIRE((p.lda | p.a_lo | p.a_s1 | p.ldw | p.w_lo) % 8 == 0 && (((uintptr_t)p.A | (uintptr_t)p.W) & 15) == 0,
                "gemm_planes: operand planes must be 16-byte aligned (strides / plane offsets multiples of 8 halves)");
    CBX_REQUIRE(!p.P || ((p.N | p.ldp | p.p_lo | p.p_s1) % 2 == 0 && ((uintptr_t)p.P & 3) == 0), "gemm_planes: plane output needs even N / strides");
    CBX_REQUIRE(p.PT || (p.pt_n0 == 0 && p.pt_T == 0), "gemm_planes: pt_n0 / pt_T without PT");
    CBX_REQUIRE((!p.C || ((long)p.M * p.ldc + p.N) * 4 < 0x7fffffffL) && (!p.R || ((long)p.M * p.ldr + p.N) * 4 < 0x7fffffffL) &&
                    (!p.P || ((long)p.M * p.ldp + p.p_lo + p.N) * 2 < 0x7fffffffL),
                "gemm_planes: one batch of an output / residual must span less than 2 GiB (32-bit buffer offsets)");
    CBX_REQUIRE(p.lda >= p.a_lo + p.Cin && p.a_lo >= 0 && p.ldw >= p.w_lo + p.K && p.w_lo >= 0,
                "gemm_planes: operand rows must hold both planes ([h | l] per row: lda >= a_lo + Cin, ldw >= w_lo + K)");
    CBX_REQUIRE(((long)p.Tin + p.taps * p.dil + 1) * p.lda * 2 < 0x7fffffffL && (long)(p.N + 256) * p.ldw * 2 < 0x7fffffffL &&
                    (long)(p.pad_left + 1) * p.lda * 2 < 0x3fffffffL,
                "gemm_planes: one batch of an operand must span less than 2 GiB (32-bit buffer offsets)");
    const int Np = p.PT ? p.pt_n0 : p.N;  // columns that go to C / P / R
    if (p.PT) {
        CBX_REQUIRE(p.P && !p.C && !p.R && p.act == CBX_ACT_NONE && p.nz1 == 1 && p.taps == 1 && p.stride == 1 && !p.lens,
                    "gemm_planes: the transposed column range serves a plain Linear with plane output (no C / R / activation / batches / conv)");
        CBX_REQUIRE(p.pt_n0 > 0 && p.pt_n0 % 256 == 0 && p.pt_n0 < p.N && p.pt_T > 0 && p.pt_T % 4 == 0 && p.M % p.pt_T == 0,
                    "gemm_planes: pt_n0 %% 256 == 0, 0 < pt_n0 < N, pt_T %% 4 == 0, M %% pt_T == 0 (got pt_n0=%d pt_T=%d M=%d N=%d)", p.pt_n0, p.pt_T, p.M, p.N);
        CBX_REQUIRE((p.pt_ld | p.pt_lo | p.pt_zs) % 4 == 0 && ((uintptr_t)p.PT & 7) == 0 && p.pt_ld >= p.pt_T && p.pt_lo > 0,
                    "gemm_planes: transposed planes need 8-byte alignment (pt_ld, pt_lo, pt_zs multiples of 4 halves)");
        CBX_REQUIRE(((long)(p.M / p.pt_T) * p.pt_zs + (long)(p.N - p.pt_n0) * p.pt_ld + p.pt_lo) * 2 < 0x7fffffffL, "gemm_planes: the transposed output must span less than 2 GiB");
    }
    CBX_REQUIRE((!p.C || p.ldc >= Np) && (!p.R || p.ldr >= Np) && (!p.P || (p.p_lo > 0 && p.ldp >= p.p_lo + Np)),
                "gemm_planes: rows must not overlap (ldc, ldr >= N; plane output rows hold [h | l]: ldp >= p_lo + N)");
    hipStream_t st = (hipStream_t)stream;
    if (p.ln_w) {  // LayerNorm of the finished row in the epilogue: the row-spanning 64 x 256 tile (one form; cbx_gemm_pl_t.tile does not apply)
        CBX_REQUIRE(p.N == 256 && p.LNP && !p.P && !p.PT && p.act == CBX_ACT_NONE, "gemm_planes: the LayerNorm epilogue needs N == 256, LNP, no P / PT / activation (N=%d)", p.N);
        CBX_REQUIRE((p.ld_lnp | p.lnp_lo | p.lnp_s1) % 2 == 0 && ((uintptr_t)p.LNP & 3) == 0 && p.lnp_lo > 0 && p.ld_lnp >= p.lnp_lo + 256 &&
                        ((long)p.M * p.ld_lnp + p.lnp_lo + 256) * 2 < 0x7fffffffL && ((uintptr_t)p.ln_w & 3) == 0,
                    "gemm_planes: LayerNorm planes need even strides, ld_lnp >= lnp_lo + 256, less than 2 GiB per batch");
        return launch_pl_act<64, 256, 2, 4, 32, 3, PL_ACT_LNF, 0>(p, st);
    }
    CBX_REQUIRE(!p.LNP, "gemm_planes: LNP without ln_w");
    const bool k64 = p.Cin % 64 == 0;
    const int df_kind = (p.P && !p.C && !p.R) ? 1 : (p.C && !p.P && !p.PT) ? 2 : 0;  // what the deferred-epilogue forms serve (else: their plain twins)
    int force = p.tile ? p.tile : g_pl_tile;  // the call's own choice first (ABI v13); the process-wide knob is a test hook
    if (force == CBX_PL_TILE_CORESIDENT) {    // one 8-wave workgroup per CU (96 KiB of LDS, <= 120 VGPRs); small grids / narrow outputs keep their forms (they never fill a CU)
        const long g128c = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nz1;
        constexpr int co_k64 = 8;  // 8 = the 128 x 128 x 64 two-stage form (128 KiB, 8 waves x 117 VGPRs): 214.5x against 211.3x for form 17 in the throughput schedule, same box (A/B hook)
        force = (g128c < 64 || p.N <= 96) ? 0 : (k64 ? co_k64 : 17);
        // ... except the wide Linear with an activation (ff1 + GELU): its deferred-epilogue form (12 waves, 168 VGPRs: nothing else fits beside it) is enough faster
        // that the throughput schedule gains from it as well (same box, A / B / A / B: 209.7 / 211.7 / 209.5 / 213.0 x, profiles/r06_s_coresident_ff1_ab.log)
        if (force && k64 && df_kind == 1 && p.act != CBX_ACT_NONE && p.N >= 512) force = 42;
        // ... and q | k | v on its loader-wave form (A / B / A / B: 218.8 / 221.3 / 219.7 / 220.8 x, profiles/r06_t_coresident_qkv_ab.log)
        else if (force && k64 && df_kind == 1 && p.N >= 512) force = 35;
#ifdef CBX_EXP_CORESIDENT_OFF  // side-library experiment (scripts/r06/r06_v.sh): every plane GEMM of the throughput schedule on its serial-schedule form
        force = 0;
#endif
    }
    // tile menu (BM x BN, waves, wave tile, BK, LDS stages); the automatic choice below is the measured one: profiles/r03_gemm_planes_tiles.log
    switch (force) {
        case 1: return launch_pl<128, 64, 4, 2, 32>(p, st);       // 8 waves 32x32, 48 KB
        case 2: return launch_pl<128, 64, 2, 2, 32>(p, st);       // 4 waves 64x32, 48 KB
        case 3: return launch_pl<128, 128, 2, 2, 32>(p, st);      // 4 waves 64x64, 64 KB
        case 4: return launch_pl<128, 128, 4, 2, 32>(p, st);      // 8 waves 32x64, 64 KB
        case 5: if (k64) return launch_pl<128, 64, 4, 2, 64>(p, st); break;    // 96 KB
        case 6: if (k64) return launch_pl<128, 64, 2, 2, 64>(p, st); break;    // 96 KB
        case 7: if (k64) return launch_pl<128, 128, 2, 2, 64>(p, st); break;   // 128 KB
        case 8: if (k64) return launch_pl<128, 128, 4, 2, 64>(p, st); break;   // 128 KB
        case 9: return launch_pl<64, 64, 2, 2, 32>(p, st);        // 4 waves 32x32, 32 KB
        case 10: return launch_pl<256, 64, 4, 2, 32>(p, st);      // 8 waves 64x32, 80 KB
        case 11: return launch_pl<256, 128, 4, 2, 32>(p, st);     // 8 waves 64x64, 96 KB
        case 12: return launch_pl<128, 128, 2, 2, 32, 3>(p, st);  // 4 waves 64x64, 3 stages, 96 KB
        case 13: return launch_pl<128, 64, 2, 2, 32, 3>(p, st);   // 4 waves 64x32, 3 stages, 72 KB
        case 14: return launch_pl<128, 128, 2, 4, 32>(p, st);     // 8 waves 64x32, 64 KB
        case 15: return launch_pl<128, 256, 2, 4, 32>(p, st);     // 8 waves 64x64, 96 KB
        case 16: return launch_pl<64, 128, 2, 2, 32>(p, st);      // 4 waves 32x64, 48 KB
        case 17: return launch_pl<128, 128, 4, 2, 32, 3>(p, st);  // 8 waves 32x64, 3 stages, 96 KB
        case 18: return launch_pl<256, 128, 4, 2, 32, 3>(p, st);  // 8 waves 64x64, 3 stages, 144 KB
        case 19: return launch_pl<128, 256, 2, 4, 32, 3>(p, st);  // 8 waves 64x64, 3 stages, 144 KB
        // loader-wave forms (8 consumer waves + 1 loader wave, one workgroup per CU)
        case 21: if (k64) return launch_pl<128, 128, 4, 2, 64, 2, 1>(p, st); break;  // 128 KB
        case 22: return launch_pl<128, 128, 2, 4, 32, 3, 1>(p, st);   // 96 KB
        case 23: return launch_pl<256, 128, 4, 2, 32, 3, 1>(p, st);   // 144 KB, 64x64 per wave
        case 24: return launch_pl<128, 128, 4, 2, 32, 2, 1>(p, st);   // 64 KB
        case 25: return launch_pl<128, 128, 2, 4, 32, 4, 1>(p, st);   // 4 stages, 128 KB
        // 16-wave workgroups (1024 threads, one per CU, 4 waves per SIMD): 25 % less operand traffic per FLOP than 128 x 128 at the same 32 x 64 wave tile
        case 26: return launch_pl<256, 128, 8, 2, 32>(p, st);      // 96 KB
        case 27: return launch_pl<256, 128, 8, 2, 32, 3>(p, st);   // 144 KB
        case 28: return launch_pl<128, 256, 4, 4, 32>(p, st);      // 96 KB
        // round 5: several loader waves (8 consumers + 2 or 4 loaders, one workgroup per CU)
        case 31: if (k64) return launch_pl<128, 128, 4, 2, 64, 2, 2>(p, st); break;   // 128 KB
        case 32: if (k64) return launch_pl<128, 128, 4, 2, 64, 2, 4>(p, st); break;   // 128 KB
        case 33: return launch_pl<128, 128, 2, 4, 32, 3, 4>(p, st);                   // 96 KB
        case 34: return launch_pl<128, 128, 2, 4, 32, 4, 4>(p, st);                   // 128 KB
        case 35: if (k64) return launch_pl<128, 128, 2, 4, 64, 2, 4>(p, st); break;   // 128 KB, 64 x 32 per wave
        case 36: return launch_pl<256, 128, 4, 2, 32, 3, 4>(p, st);                   // 144 KB, 64 x 64 per wave
        case 37: return launch_pl<128, 128, 4, 2, 32, 4, 2>(p, st);                   // 128 KB
        // round 6: the deferred epilogue (a finished tile's epilogue rides on the next tile's K loop)
        case 41:  // form 32 + DF
            if (k64 && df_kind == 1) return launch_pl<128, 128, 4, 2, 64, 2, 4, 1>(p, st);
            if (k64 && df_kind == 2) return launch_pl<128, 128, 4, 2, 64, 2, 4, 2>(p, st);
            if (k64) return launch_pl<128, 128, 4, 2, 64, 2, 4>(p, st);
            break;
        case 42:  // form 35 (64 x 32 per wave) + DF
            if (k64 && df_kind == 1) return launch_pl<128, 128, 2, 4, 64, 2, 4, 1>(p, st);
            if (k64 && df_kind == 2) return launch_pl<128, 128, 2, 4, 64, 2, 4, 2>(p, st);
            if (k64) return launch_pl<128, 128, 2, 4, 64, 2, 4>(p, st);
            break;
        // round 6: deeper rings / wider K tiles for the 64 x 64 tile of small grids (batch 1: one K tile in flight is a DMA round trip per K tile)
        case 43: return launch_pl<64, 64, 2, 2, 32, 3>(p, st);                        // 48 KB
        case 44: if (k64) return launch_pl<64, 64, 2, 2, 64, 2>(p, st); break;        // 64 KB
        case 45: if (k64) return launch_pl<64, 64, 2, 2, 64, 3>(p, st); break;        // 96 KB
        default: break;
    }
    // automatic choice (rows 16 x T 1000; profiles/r03_bench_planes_tiles.log, round 5: profiles/r05_bench_planes_loader_waves.log): 8 consumer waves
    // everywhere.  The loader-wave forms (128 x 128 x 64, one workgroup per CU) win wherever K % 64 == 0 -- with FOUR loader waves (one per SIMD, each
    // streaming one of {A.h, W.h, A.l, W.l}) since round 5: 2-5 % under the one-loader form on every CFM shape, and now also ahead of the symmetric two-
    // workgroup form for the GELU epilogue (ff1: 43.7 against 46.1 us); N <= 96 takes half-width tiles, small grids 64 x 64.
    const long g128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nz1;
    // part-filled chip (batch 1 - 4: at most 128 tiles of 128 x 128): 64 x 64 tiles, with 64-wide K tiles where the shape allows (half the barriers and DMA round trips of a
    // K walk that is latency-bound at one workgroup per CU) -- round 6, profiles/r06_ab_plane_gemm_small_grids.log: ff2 at 4 rows 18.6 -> 13.1 us, at 2 rows 13.6 -> 12.5;
    // the crossover with the 128 x 128 forms lies between 126 and 134 such tiles for the N = 256 shapes and at 128 for the wide ones
    auto small = [&]() { return k64 ? launch_pl<64, 64, 2, 2, 64, 2>(p, st) : launch_pl<64, 64, 2, 2, 32>(p, st); };
    if (g128 < 64) return small();
    if (p.N <= 96) return launch_pl<128, 64, 4, 2, 32>(p, st);
    if (g128 <= 128) return small();
    // round 6 (INTERLEAVED rounds, profiles/r06_p_plane_gemm_forms_interleaved.log): plane outputs of the wide Linears take 64 x 32 per wave; with an activation (ff1: GELU)
    // the deferred epilogue, whose VALU rides in the next tile's MFMA shadow (41.4 against 48.4 us, at 64 rows 182 against 217); without one (q | k | v) its plain twin --
    // there the epilogue is stores, which the deferred form only moves into the DMA-bound K loop (50.9 against 53.7 us at 16 rows, 272 against 255 at 64)
#ifndef CBX_EXP_NO_DF  // (side-library experiment, scripts/r06/r06_u.sh: the round-5 choice)
    if (p.N >= 512 && k64 && df_kind == 1)
        return p.act != CBX_ACT_NONE ? launch_pl<128, 128, 2, 4, 64, 2, 4, 1>(p, st) : launch_pl<128, 128, 2, 4, 64, 2, 4>(p, st);
#endif
    if (p.N >= 512) return k64 ? launch_pl<128, 128, 4, 2, 64, 2, 4>(p, st) : launch_pl<128, 128, 2, 4, 32>(p, st);
    if (k64 && p.K >= 512 && df_kind == 2 && g128 > 256) return launch_pl<128, 128, 2, 4, 64, 2, 4, 2>(p, st);  // more tiles than CUs (64 rows): 61.4 against 68.7 us
    if (k64 && p.K >= 512) return launch_pl<128, 128, 4, 2, 64, 2, 4>(p, st);
    return launch_pl<128, 128, 4, 2, 32>(p, st);
}

extern "C" int cbx_split_planes_f32(const float* x, void* planes, long rows, int C, long ldx, long ldp, long p_lo, void* stream) {
    CBX_REQUIRE(x && planes && C % 4 == 0 && ldx % 4 == 0 && ldp % 4 == 0 && p_lo % 4 == 0, "split_planes: C, ldx, ldp, p_lo must be multiples of 4");
    CBX_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)planes) & 7) == 0, "split_planes: alignment");
    if (rows <= 0) return 0;
    const long n = rows * (C / 4);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<_Float16*>(planes), rows, C / 4, ldx, ldp, p_lo, cbx_range_flag());
    return cbx_check_launch("split_planes");
}
