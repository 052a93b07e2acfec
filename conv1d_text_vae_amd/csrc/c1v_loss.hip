// a8 (cosine-softmax cross-entropy vs the vocabulary matrix) and a10-ii (nearest vocabulary row) of libc1dvae.
//
// The fused bf16 kernel `vocab_stream_kernel` is the dominant kernel of a training step: per workgroup it keeps
// 128 rows of the (normalised, pre-scaled) decoder output in registers as MFMA B operands, streams the vocabulary
// matrix through LDS in 64-row tiles (LDS-DMA, double buffered) and per tile computes
//     S^T = W_tile . Yhat^T            (v_mfma_f32_32x32x16_bf16, vocabulary row on the accumulator ROW,
//                                       sample row on the LANE)
//     E   = exp2(S^T)                  (logits are bounded: |cos| <= 1, tau = 0.1 => no running max needed)
//     U^T += W_tile^T . E              (the accumulator tile IS the next MFMA's B operand; W^T fragments come from
//                                       the same LDS image through ds_read_b64_tr_b16)
// so the [M, N] logits never exist in memory.  Column `ones_col` of W holds 1.0, which makes row `ones_col` of
// U^T the softmax denominator for free.  Modes: ACCUM (training: U^T), SUMEXP (validation: only the denominator),
// ARGMAX (predict: fused per-row arg-max, lowest index wins ties).
#include "c1v_common.h"
#include <stdlib.h>
#include <type_traits>
#include "c1v_lossred.h"

// ------------------------------------------------------------------------------------------------------
// l2 normalise rows: yhat = y * rsqrt(max(sum y^2, 1e-12)) * qscale      (K.l2_normalize, reference :1195)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ y, int M, int V, int64_t ldy,
                                                          T* __restrict__ yhat, int kp, float* __restrict__ rnorm, float qscale,
                                                          const int* __restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // live-row compaction (c1v_live_rows): a row without a target takes no part in the loss -- it is not normalised at
    // all; the others land at their compact position
    const int orow = pos ? pos[row] : row;
    if (orow < 0) return;
    const float* yr = y + (int64_t)row * ldy;
    // rows of at most 512 columns: every lane requests its columns (lane + 64 r, clamped) before the first is used, keeps
    // them in registers and never reads the row a second time (the plain two-pass loop paid one round trip per 64 columns)
    constexpr int R = 8;
    if (kp <= 64 * R) {
        float v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { const int c = lane + 64 * r; v[r] = yr[c < V ? c : V - 1]; }
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) if (lane + 64 * r < V) ss += v[r] * v[r];
        ss = wave_sum(ss);
        const float rr = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        if (lane == 0 && rnorm) rnorm[row] = rr;
        const float sc = rr * qscale;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c = lane + 64 * r;
            if (c < kp) ElemTraits<T>::store(yhat, (int64_t)orow * kp + c, c < V ? v[r] * sc : 0.0f);
        }
        return;
    }
    float ss = 0.f;
    for (int c = lane; c < V; c += 64) { const float v = yr[c]; ss += v * v; }
    ss = wave_sum(ss);
    const float r = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    if (lane == 0 && rnorm) rnorm[row] = r;
    const float sc = r * qscale;
    for (int c = lane; c < kp; c += 64) ElemTraits<T>::store(yhat, (int64_t)orow * kp + c, c < V ? yr[c] * sc : 0.0f);
}

__global__ __launch_bounds__(1024) void live_rows_kernel(const int* __restrict__ ids, int M, int items, int vocab_tiles, int max_splits,
                                                         int* __restrict__ pos, int* __restrict__ ctl, int* __restrict__ rows, int* __restrict__ cids) {
    live_rows_body(ids, M, items, vocab_tiles, max_splits, pos, ctl, rows, cids);
}

// ------------------------------------------------------------------------------------------------------
// fused vocabulary stream kernel (bf16)
// ------------------------------------------------------------------------------------------------------
enum { VS_ACCUM = 0, VS_SUMEXP = 1, VS_ARGMAX = 2 };
// Two round-2 variants of the ACCUM tile, both parity-green, both measured at C2 and C4 and neither faster than the plain
// form (47.2-48.8 us at C2 either way; stamps in DESIGN.md section 5), kept as diagnostic builds:
//   -DVS_INTERLEAVE=1  block 1's exp2 / bf16 conversion rides in the MFMA gaps of GEMM2 on block 0 (one v_exp_f32 or two
//                      v_cvt_pk per gap, which profiles/microbench/mfma_chain.hip shows to be free next to a 32-cycle MFMA)
//   -DVS_REGFEED=1|2   the LDS ring is fed through registers (buffer_load_dwordx4 + ds_write_b128, one piece per MFMA gap)
//                      instead of LDS-DMA requests; 2 = the reload follows the ds_write at once (a whole tile to arrive)
#ifndef VS_INTERLEAVE
#define VS_INTERLEAVE 0
#endif
#ifndef VS_REGFEED
#define VS_REGFEED 0
#endif
#ifdef VS_STAMP
// diagnostic build only: per-segment cycle sums of wave 0 of every workgroup (guide: "In-kernel stamps")
__device__ unsigned long long vs_dbg[8 * 4096];
#define VS_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    seg[i] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int c1v_debug_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vs_dbg), sizeof(unsigned long long) * n);
}
#else
#define VS_T(i) do {} while (0)
#endif
//   -DVS_PROLOGUE_OVERLAP=1  (round 3) the prologue's requests ordered tile 0 | Yhat | tile 1, a counted wait for tile 0 only and
//                      the first tile peeled so that its MFMAs wait for Yhat k-step by k-step: parity-green, measured at C2
//                      49.8 us against 48.8 (the first tile's barrier waits for everything half a tile later anyway)
#ifndef VS_XCD_ROWS
#define VS_XCD_ROWS 1
#endif
#ifndef VS_PROLOGUE_OVERLAP
#define VS_PROLOGUE_OVERLAP 0
#endif
constexpr int VS_BN = 64;     // vocabulary rows per LDS tile
constexpr int VS_BM = VS_BM_ROWS;    // sample rows per workgroup (4 waves x 32)
constexpr int VS_STAGES = 3;  // LDS ring of vocabulary tiles

template <int CB> struct VsCfg {
    static constexpr int KP = CB * 32;
    static constexpr int ROWB = KP * 2;                                  // data bytes per row
    static constexpr int PITCH = ROWB + ((CB % 2 == 0) ? 64 : 128);      // == 64 or 192 (mod 256)
    static constexpr int CPR = PITCH / 16;                               // physical 16-byte chunks per row
    static constexpr int DCH = ROWB / 16;                                // data chunks per row
    static constexpr int STAGE = VS_BN * PITCH;
    static constexpr int PIECES = STAGE / 1024;                          // 1 KiB LDS-DMA pieces per tile
    static constexpr int PPW = (PIECES + 3) / 4;                         // pieces per wave
    static_assert(STAGE % 1024 == 0, "tile must be whole LDS-DMA pieces");
    static_assert(PIECES % 4 == 0, "every wave issues the same number of pieces (counted vmcnt waits rely on it)");
    static_assert(PITCH % 256 == 64 || PITCH % 256 == 192, "pitch keeps the transposed reads conflict-free");
};

// byte offset of 16-byte chunk `ch` of row r inside a tile image (chunk index XOR-swizzled in its low 2 bits)
template <int CB> __device__ __forceinline__ int vs_off(int r, int ch) { return r * VsCfg<CB>::PITCH + ((ch ^ ((r >> 2) & 3)) << 4); }

template <int CB, int MODE, int MB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void vocab_stream_kernel(const bf16_raw* __restrict__ yhat, int M, const bf16_raw* __restrict__ w, int N, int n_base,
                         int tiles_total, int tiles_per_split, float* __restrict__ ut, int ut_bf16, int m_pad, int ones_col,
                         float* __restrict__ part_score, int* __restrict__ part_idx, const int* __restrict__ ctl) {
    using Cfg = VsCfg<CB>;
    constexpr int KS = CB * 2;                     // 16-wide k steps of GEMM1
    // Sample rows per wave = 32 MB.  MB = 1 with the U accumulators of ACCUM mode (160 registers); the GEMM1-only modes also
    // come with MB = 2, where every vocabulary fragment read from LDS feeds two MFMAs (LDS bandwidth co-limits the kernel:
    // all four waves read the whole tile) -- used when there are enough rows to fill the chip with 256-row workgroups
    static_assert(MB == 1 || MODE != VS_ACCUM, "ACCUM has no registers for a second m-block");
    constexpr int WG_ROWS = VS_BM * MB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VS_STAMP
    unsigned long long t_entry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, split = blockIdx.y;       // row tile, vocabulary split
    if (ctl) {
        // compacted rows (live_rows_kernel): the grid is a pool of workgroups, dealt to (row tile, split) on the device
        // (A split-major deal that keeps the workgroups of one XCD on one or two vocabulary splits -- so that each L2 fetches
        // 2/splits of W instead of all of it -- was measured at C2: 1 % SLOWER, 49.0 against 48.5 us; W is L2/MALL-resident
        // either way and 32 CUs asking for the same lines at the same time is worse than 5 of them.)
        const int live = ctl[0], splits = ctl[2], full = ctl[4];
        const int item = blockIdx.x + gridDim.x * blockIdx.y;
        if (item < full) {                          // a whole round's row tile: the whole vocabulary, slab 0
            bx = item; split = 0; tiles_per_split = tiles_total;
        } else {                                    // the remainder, cut into ctl[2] splits each
            const int r = item - full;
#if VS_XCD_ROWS
            // workgroups go to the XCDs round-robin (XCD = id % 8): the splits of ONE row tile on ONE XCD, so that its L2
            // fetches the tile's 80 KiB of Yhat once for all of them (every workgroup of the launch loads its rows at the
            // same time: the prologue runs at the ~11 B/clk a CU gets in such a burst).  Whole tiles per XCD first; the
            // tiles that are left over take the left-over slots of all XCDs.
            const int pool = (int)(gridDim.x * gridDim.y) - full;
            const int xcd = r & 7, slot = r >> 3, spx = pool >> 3, tpx = spx / splits;
            if ((pool & 7) == 0 && tpx > 0) {
                if (slot < tpx * splits) { bx = (slot / splits) * 8 + xcd; split = slot % splits; }
                else { const int lo = (slot - tpx * splits) * 8 + xcd; bx = 8 * tpx + lo / splits; split = lo % splits; }
                bx += full;
            } else
#endif
            { bx = r / splits; split = r - bx * splits; bx += full; }
            tiles_per_split = ctl[3];
        }
        if (bx * WG_ROWS >= live) return;           // whole workgroup, before any barrier
        M = live;
    }
    const int m_row0 = bx * WG_ROWS + wave * 32 * MB + l31;      // this lane's row of m-block 0 (m-block mb: + 32 mb)
    const int t_begin = split * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;

    // LDS-DMA source offsets of this lane for the pieces of its wave (bytes from the tile's first row); every wave owns
    // PPW pieces (PIECES % 4 == 0), so no piece is conditional and counted vmcnt waits are the same in all waves
    int src_off[Cfg::PPW];
#pragma unroll
    for (int i = 0; i < Cfg::PPW; ++i) {
        const int piece = wave + 4 * i;
        const int L = piece * 64 + lane;
        const int r = L / Cfg::CPR, chp = L - r * Cfg::CPR;
        const int ch = chp ^ ((r >> 2) & 3);
        src_off[i] = (chp < Cfg::DCH) ? r * Cfg::ROWB + ch * 16 : 0;
    }
    auto stage_piece = [&](int t, int stage, int i) {
        const char* tile = (const char*)w + (int64_t)t * VS_BN * Cfg::ROWB;
        c1v_glds16_so(tile, (uint32_t)src_off[i], c1v_lds_addr(smem + stage * Cfg::STAGE + (wave + 4 * i) * 1024));
    };
    auto stage_tile = [&](int t, int stage) {
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) stage_piece(t, stage, i);
    };

    // three LDS stages: tile t+2 is requested when tile t starts, so a request has two tile times to arrive (with two
    // stages the L2/fabric latency of a 44 KiB request, not the MFMA work, set the tile time)
    // ACCUM feeds the ring through REGISTERS instead (REGFEED): an LDS-DMA request costs the issuing wave 60-100 cycles
    // wherever it is placed (one per MFMA gap was measured too: slower than the burst), and with one wave per SIMD and no
    // registers for a feeder wave that was ~20 % of the tile.  A plain global_load_dwordx4 and, one tile later, a
    // ds_write_b128 per piece issue in the shadow of the MFMAs: piece i of tile t+2 is loaded behind MFMA i of GEMM2(t)
    // and written to its stage behind one of the last MFMAs of GEMM1(t+1), in front of the barrier that publishes it.
    constexpr bool REGFEED = MODE == VS_ACCUM && VS_REGFEED != 0;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));          // (a plain vector: no aggregate copies)
    u32x4_t R[REGFEED ? Cfg::PPW : 1];
    // (macros, not lambdas: behind a second level of closures the array stayed in scratch memory)
    // (buffer loads: SGPR descriptor + SGPR tile offset + the lane's 32-bit offset -- no 64-bit address arithmetic per piece)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
#define VS_LOAD_PIECE(tb_, i_) R[i_] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, src_off[i_], (tb_), 0)
    auto tile_base = [&](int t) -> int { return __builtin_amdgcn_readfirstlane(t * (VS_BN * Cfg::ROWB)); };   // bytes, < 2^31 (host check)
#define VS_WRITE_PIECE(stage_, i_) *(u32x4_t*)(smem + (stage_) * Cfg::STAGE + (wave + 4 * (i_)) * 1024 + lane * 16) = R[i_]
    // Order of the prologue's requests (they complete in order): tile 0, the Yhat fragments, tile 1.  The wait in front of
    // the first barrier covers tile 0 only (this wave's pieces); the first tile -- a copy of the loop body of its own, below,
    // so that the compiler counts its waits per k-step -- starts when its first k-steps of Yhat are there, and the rest of
    // Yhat and tile 1 (most of the 42 KiB a wave asks for, fetched at the ~11 B/clk a CU gets while every workgroup of the
    // launch does the same) arrive under its MFMAs.  ACCUM only: the other modes' first barrier sits inside the tile.
    constexpr bool OVERLAP = MODE == VS_ACCUM && VS_PROLOGUE_OVERLAP != 0;
    if (t_begin < t_end) stage_tile(t_begin, 0);
    if constexpr (!REGFEED && !OVERLAP) {
        if (t_begin + 1 < t_end) stage_tile(t_begin + 1, 1);
    }

    // this wave's 32 sample rows as MFMA B operands: B[k = 8h + j][col = l31] = yhat[m_row][16 s + 8h + j]
    bf16x8_t yf[MB][KS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m_row = m_row0 + 32 * mb;
        const uint4* yp = (const uint4*)(yhat + (int64_t)(m_row < M ? m_row : 0) * Cfg::KP) + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // (OVERLAP: unconditional -- a lane behind the last row reads row 0 again and computes a copy of its results, which
            // nothing consumes: every sample row is a column of its own in S, E and U.  A load under `if` is a basic block of its
            // own, and the compiler then waits for ALL of them in front of the first MFMA instead of k-step by k-step)
            uint4 v;
            if constexpr (OVERLAP) v = yp[2 * s];
            else v = (m_row < M) ? yp[2 * s] : make_uint4(0, 0, 0, 0);
            yf[mb][s] = __builtin_bit_cast(bf16x8_t, v);
        }
    }

    if constexpr (REGFEED) {
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) VS_LOAD_PIECE(tile_base(t_begin + 1 < t_end ? t_begin + 1 : t_end - 1), i);
    } else if constexpr (OVERLAP) {
        // (a last tile is requested twice rather than under a branch: the counted wait below assumes PPW requests)
        if (t_begin < t_end) stage_tile(t_begin + 1 < t_end ? t_begin + 1 : t_begin, 1);
    }

    f32x16_t U[MODE == VS_ACCUM ? CB : 1];
#pragma unroll
    for (int c = 0; c < (MODE == VS_ACCUM ? CB : 1); ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) U[c][r] = 0.f;
    float sumexp[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) sumexp[mb] = 0.f;
    float best[MB];
    int best_idx[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) { best[mb] = -3.0e38f; best_idx[mb] = 0x7fffffff; }

    // Both requests and the yf loads (one-time cost).  The BUILTIN form on purpose: the compiler's own bookkeeping must
    // see that the yf loads are complete, or it keeps a `vmcnt(0)` in front of the first MFMA of EVERY tile (an asm wait
    // is invisible to it), which waits for the LDS-DMA request issued a few instructions earlier.
    if constexpr (OVERLAP) {
        constexpr int VMW = KS * MB + Cfg::PPW;            // requests behind tile 0's: Yhat fragments + tile 1
        static_assert(VMW < 64, "vmcnt is a 6-bit counter");
        __builtin_amdgcn_s_waitcnt(((VMW >> 4) << 14) | 0x0F70 | (VMW & 15));
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt / lgkmcnt untouched
    }
    asm volatile("" ::: "memory");
    __syncthreads();

    // Software pipeline of one tile (one wave per SIMD, so nothing else hides LDS latency): operand fragments are fetched
    // one BATCH ahead into a second register buffer.  sched_barrier(0) pins "loads of batch q+1, then MFMAs of batch q"
    // (the compiler otherwise re-serialises every ds_read behind an lgkmcnt(0) in front of its MFMA, which tripled the
    // tile time).
    // GEMM1 batch: KB k-steps of BOTH 32-row blocks of the tile, issued alternately -- consecutive MFMAs never accumulate
    // into the same registers.  Measured on gfx950 (profiles/microbench/mfma_chain.hip, s_memtime cycles): a chain of dependent
    // 32x32x16 MFMAs issues every 44 cycles, two or more independent chains every 32; behind each MFMA the wave may issue 8 cycles
    // of VALU work for free (one v_exp_f32, or two v_fma / v_cvt_pk), anything more delays the next MFMA cycle for cycle -- with
    // one or two waves per SIMD alike.  The VALU stages simply follow GEMM1 (see VS_INTERLEAVE above for the measured alternative).
    constexpr int KB = 2, G1 = 2 * KB;                  // k-steps / fragments per batch
    constexpr int Q1 = KS / KB, HB = (Q1 + 1) / 2;      // batches; the GEMM1-only modes put their barrier behind batch HB-1
    static_assert(KS % KB == 0, "k-steps come in pairs");
    constexpr int CG = (CB % 2 == 0) ? 2 : 1;           // GEMM2: cb blocks per batch (2 k-steps each)
    constexpr int NB2 = CB / CG, Q2 = 2 * NB2;
    const int tq = (lane & 15) >> 2, tpq = lane & 3, tcg = (lane >> 4) & 1;

#ifdef VS_STAMP
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
    const unsigned long long tstart = tprev;
#endif
    // ACCUM mode runs a tile as  GEMM1(t) | exp2, bf16 | wait + barrier | GEMM2(t)  with ONE barrier, in the middle of the tile:
    //   * it publishes tile t+1 (requested during GEMM2 of tile t-1) and proves that every wave is done with tile t-1, whose
    //     stage the request for tile t+2 -- sent right after it, behind the GEMM2 batches -- overwrites;
    //   * nothing separates the end of GEMM2(t) from GEMM1(t+1), so the first fragments of tile t+1 are fetched behind the
    //     last GEMM2 batch of tile t.
    // The other modes (no GEMM2) put the same single barrier in the middle of GEMM1.  In-kernel stamps (VS_STAMP build) of
    // the ACCUM form at C2: GEMM1 and the second half of GEMM2 run at the MFMA rate (~37 cycles each); the LDS-DMA requests
    // and their LDS writes cost ~20 % of the tile, the exp2 / bf16 stage ~8 %; the prologue (first two tiles and the Yhat
    // fragments, every workgroup at once) is ~10 % of the kernel, the U write-out ~4 %.  A form with wave PAIRS (each LDS fragment
    // feeding two MFMAs, half the LDS reads) was built and measured: no faster, so LDS read bandwidth is not the limit.
    // LDS addresses of the fragment reads = (stage base + a lane constant) + a COMPILE-TIME offset, so that every ds_read
    // carries its offset as an immediate and a tile needs four address additions in all (the swizzle only touches the low
    // two chunk bits and depends on the row through (row >> 2) & 3, which the +32 / +16 row steps of the fragments leave
    // alone; written with vs_off() per read, the compiler spent ~130 VALU instructions per tile on addresses, and VALU
    // instructions are not free next to MFMAs):
    //   GEMM1 fragment (nb, s):      vs_off(nb*32 + l31, 2s + h)        = g1off[s & 1] + nb*32*PITCH + (s >> 1)*64
    //   GEMM2 fragment (nb, ks, cb): vs_off(nb*32 + 16ks + R [+8], 4cb + c0) + sub = g2off[lo|hi] + (nb*32 + 16ks)*PITCH + cb*64
    int g1off[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) g1off[par] = vs_off<CB>(l31, 2 * par + h);
    const int g2off_lo = vs_off<CB>(4 * h + tq, 2 * tcg + (tpq >> 1)) + 8 * (tpq & 1);
    const int g2off_hi = vs_off<CB>(4 * h + tq + 8, 2 * tcg + (tpq >> 1)) + 8 * (tpq & 1);
    auto ld1_at = [&](const char* base, int nb, int s) -> uint4 {
#if defined(VS_ABLATE) && (VS_ABLATE & 32)
        return make_uint4(lane + s, nb, 0x3f803f80u, 0x3f803f80u);          // diagnostic: GEMM1 without its LDS reads
#else
        return *(const uint4*)(base + g1off[s & 1] + nb * 32 * Cfg::PITCH + (s >> 1) * 64);
#endif
    };
    uint4 a1[2][G1];
    if (t_begin < t_end) {
#pragma unroll
        for (int f = 0; f < G1; ++f) a1[0][f] = ld1_at(smem, f & 1, f >> 1);
    }
    // VALU stage on accumulator regs [r0, r1) of block nb of the tile whose first vocabulary row is pn0 (tile index pt):
    // exp2 (+ tail mask) / running sum / running arg-max.  tailc: the tile hangs over the end of the vocabulary.
    auto post = [&](f32x16_t (&S)[MB][2], int nb, int r0, int r1, int pn0, int mb0, int mb1, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
#pragma unroll
        for (int mb = mb0; mb < mb1; ++mb) {
            if constexpr (MODE == VS_ARGMAX) {
                // Running arg-max per lane: the slice's maximum first (a few v_max), and only when it beats the running
                // best of some lane -- ever rarer as the stream goes on -- the element-by-element search that also yields
                // the row (in ascending row order with a strict compare: the lowest row wins ties).  Tracking the row of
                // every logit cost ~8 VALU instructions each, as much issue time as the MFMAs of the tile.
                float v[16];
#pragma unroll
                for (int r = r0; r < r1; ++r) {
                    v[r] = S[mb][nb][r];
                    if constexpr (TAIL) { const int n = pn0 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; if (n >= N) v[r] = -3.0e38f; }
                }
                float m = v[r0];
#pragma unroll
                for (int r = r0 + 1; r < r1; ++r) m = fmaxf(m, v[r]);
                if (__builtin_amdgcn_ballot_w64(m > best[mb]) != 0) {
#pragma unroll
                    for (int r = r0; r < r1; ++r) {
                        const int n = pn0 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (v[r] > best[mb]) { best[mb] = v[r]; best_idx[mb] = n; }
                    }
                }
            } else {
#pragma unroll
                for (int r = r0; r < r1; ++r) {
#if defined(VS_ABLATE) && (VS_ABLATE & 1)
                    float e = S[mb][nb][r];
#else
                    float e = __builtin_amdgcn_exp2f(S[mb][nb][r]);
#endif
                    if constexpr (MODE == VS_SUMEXP) {
                        // (ACCUM needs no tail mask: rows >= N of W are all-zero, ones column included, so whatever
                        // exp2 gives for them is multiplied by zero in GEMM2)
                        if constexpr (TAIL) { const int n = pn0 + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; if (n >= N) e = 0.f; }
                        sumexp[mb] += e;
                    }
                    S[mb][nb][r] = e;
                }
            }
        }
    };
    auto tile_body = [&](int t, auto tailc) {
        VS_T(7);
        const int stage = (t - t_begin) % VS_STAGES;
#if defined(VS_ABLATE) && (VS_ABLATE & 8)
        const char* sw = smem;
#else
        const bool more = t + 2 < t_end;            // wave-uniform: tile t+2 is requested inside GEMM1 below
        const char* sw = smem + stage * Cfg::STAGE;
#endif
        const int n0 = t * VS_BN;
        const int tb2 = REGFEED ? tile_base(t + 2 < t_end ? t + 2 : t_end - 1) : 0;

        auto ld1 = [&](int nb, int s) -> uint4 { return ld1_at(sw, nb, s); };
        // W^T fragment of GEMM2: k slot (h, j) of step (nb, ks) is vocabulary row nb*32 + 16ks + 8(j>>2) + 4h + (j&3),
        // exactly the accumulator rows held in regs 8ks..8ks+7 of S[nb]
        auto ld2 = [&](int nb, int ks, int cb) -> uint4 {
#if defined(VS_ABLATE) && (VS_ABLATE & 16)
            return make_uint4(lane + ks, nb + cb, 0x3f803f80u, 0x3f803f80u);     // diagnostic: GEMM2 without its LDS reads
#endif
            const int imm = (nb * 32 + 16 * ks) * Cfg::PITCH + cb * 64;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, sw + g2off_lo + imm));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, sw + g2off_hi + imm));
            return make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y,
                              __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
        };

        // GEMM1 accumulators: the two 32-row blocks of the vocabulary tile (never cleared: the first MFMA of every chain takes a
        // constant-zero C operand)
        f32x16_t S[MB][2];
        f32x16_t zero16;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
        uint4 a2[3][CG * 2];
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            const int s0 = q * KB;
            if (q + 1 < Q1) {
#pragma unroll
                for (int f = 0; f < G1; ++f) a1[(q + 1) & 1][f] = ld1(f & 1, s0 + KB + (f >> 1));
            } else if constexpr (MODE == VS_ACCUM) {
#pragma unroll
                for (int qq = 0; qq < 2 && qq < Q2; ++qq)
#pragma unroll
                    for (int c = 0; c < CG; ++c)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) a2[qq][c * 2 + ks] = ld2(qq / NB2, ks, (qq % NB2) * CG + c);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < G1; ++f) {
#if defined(VS_ABLATE) && (VS_ABLATE & 4)
                if (f == 0 && s0 == 0)
#endif
                {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        S[mb][f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1[q & 1][f]), yf[mb][s0 + (f >> 1)],
                                                                               (q == 0 && f < 2) ? zero16 : S[mb][f & 1], 0, 0, 0);
                }
                if constexpr (REGFEED) {
                    // tile t+1 from registers to its stage (dead since the barrier of tile t-1), one piece per gap, behind the last
                    // PPW MFMAs of GEMM1 (on the last tile this writes a stale copy into a dead stage: no branch)
                    constexpr int G0 = Q1 * G1 - Cfg::PPW;
                    const int g = q * G1 + f;
                    if (g >= G0) {
                        VS_WRITE_PIECE((stage + 1) % VS_STAGES, g - G0);
                        if (VS_REGFEED == 2) VS_LOAD_PIECE(tb2, g - G0);      // reload at once: a whole tile time to arrive
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#if !(defined(VS_ABLATE) && (VS_ABLATE & 8))
            if constexpr (MODE != VS_ACCUM) {
                // GEMM1-only modes: the tile's barrier sits behind batch HB-1, the request for tile t+2 behind a later batch
                // (a different one per wave where there are enough), the first fragments of tile t+1 behind the last batch
                if (q == HB - 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    VS_T(4);
                    __syncthreads();
                    VS_T(5);
                }
                constexpr int QD = HB < Q1 ? HB : Q1 - 1;                  // first batch behind the barrier (Q1 == 1: the batch itself)
                if ((Q1 - QD >= 4 ? q == QD + wave : q == QD) && more) {
#pragma unroll
                    for (int i = 0; i < Cfg::PPW; ++i) stage_piece(t + 2, (stage + 2) % VS_STAGES, i);
                }
                if (q == Q1 - 1 && t + 1 < t_end) {
                    const char* swn1 = smem + ((t + 1 - t_begin) % VS_STAGES) * Cfg::STAGE;
#pragma unroll
                    for (int f = 0; f < G1; ++f) a1[0][f] = ld1_at(swn1, f & 1, f >> 1);
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (q == HB - 1) VS_T(0);
            if (q == Q1 - 1) VS_T(1);
        }
        // VALU stage (VS_INTERLEAVE builds: block 0 only, block 1 rides in the MFMA gaps of GEMM2 on block 0)
        post(S, 0, 0, 16, n0, 0, MB, tailc);
        if constexpr (MODE != VS_ACCUM || VS_INTERLEAVE == 0) post(S, 1, 0, 16, n0, 0, MB, tailc);

        if constexpr (MODE == VS_ACCUM) {
            bf16x8_t pf[2][2];
#pragma unroll
            for (int nb = 0; nb < (VS_INTERLEAVE ? 1 : 2); ++nb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[nb][ks][j] = (__bf16)S[0][nb][8 * ks + j];
            // block 1 in 20 units of <= 8 issue cycles: units 0..15 = exp2 of accumulator register u, units 16..19 = conversion
            // of four values to bf16; GEMM2 on block 0 has 2 CB gaps, gap g takes units [g UPG, (g+1) UPG)
            // (asm volatile: as builtins the compiler sinks all of it to the block of first use, behind the last batch)
            constexpr int UPG = (20 + 2 * CB - 1) / (2 * CB);
            uint32_t pk1[8];
            auto gap_units = [&](int g) {
#pragma unroll
                for (int u = g * UPG; u < (g + 1) * UPG && u < 20; ++u) {
                    if (u < 16) {
                        asm volatile("v_exp_f32 %0, %0" : "+v"(S[0][1][u]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 4 * (u - 16) + 2 * e;
                            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk1[r >> 1]) : "v"(S[0][1][r]), "v"(S[0][1][r + 1]));
                        }
                    }
                }
            };
#if !(defined(VS_ABLATE) && (VS_ABLATE & 8))
            // the tile's only barrier: tile t+1 (this wave's pieces: everything it has in flight) is in LDS, tile t-1 is dead
            // (register feed: the pieces went to LDS with ds_write, which the barrier's fence waits for)
            if constexpr (!REGFEED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            VS_T(4);
            __syncthreads();
            VS_T(5);
            const char* swn = smem + ((t + 1 - t_begin) % VS_STAGES) * Cfg::STAGE;
#endif
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                const int nb = q / NB2, cb0 = (q % NB2) * CG;
                if (q + 2 < Q2) {
                    const int nb2 = (q + 2) / NB2, cb2 = ((q + 2) % NB2) * CG;
#pragma unroll
                    for (int c = 0; c < CG; ++c)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) a2[(q + 2) % 3][c * 2 + ks] = ld2(nb2, ks, cb2 + c);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (k-step outside, column block inside: consecutive MFMAs accumulate into different registers)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
#if defined(VS_ABLATE) && (VS_ABLATE & 2)
                        if (cb0 + c == 0)
#endif
                        U[cb0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a2[q % 3][c * 2 + ks]), pf[nb][ks],
                                                                             U[cb0 + c], 0, 0, 0);
                        if constexpr (VS_INTERLEAVE != 0) {
                            if (nb == 0) {                       // (compile-time: q is an unrolled index)
                                gap_units((q % NB2) * 2 * CG + ks * CG + c);
                                if (q == NB2 - 1 && ks == 1 && c == CG - 1) {
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int k2 = 0; k2 < 2; ++k2)
                                        pf[1][k2] = __builtin_bit_cast(bf16x8_t, make_uint4(pk1[4 * k2], pk1[4 * k2 + 1], pk1[4 * k2 + 2], pk1[4 * k2 + 3]));
                                }
                            }
                        }
#if !(defined(VS_ABLATE) && (VS_ABLATE & 8))
                        if constexpr (REGFEED) {
                            // tile t+2 (clamped into the split: no branch) -> registers, one piece per MFMA gap
                            const int g = q * 2 * CG + ks * CG + c;
                            if (VS_REGFEED == 1 && g < Cfg::PPW) VS_LOAD_PIECE(tb2, g);
                        }
#endif
                        if constexpr (VS_INTERLEAVE != 0 || REGFEED) __builtin_amdgcn_sched_barrier(0);
                    }
#if !(defined(VS_ABLATE) && (VS_ABLATE & 8))
                // request for tile t+2: every wave sends all its pieces behind one batch, a different batch per wave
                if (!REGFEED && (Q2 >= 8 ? q == 2 * wave : q == 0) && more) {
#pragma unroll
                    for (int i = 0; i < Cfg::PPW; ++i) stage_piece(t + 2, (stage + 2) % VS_STAGES, i);
                }
                // first GEMM1 fragments of tile t+1, behind the last batch
                if (q == Q2 - 1 && t + 1 < t_end) {
#pragma unroll
                    for (int f = 0; f < G1; ++f) a1[0][f] = ld1_at(swn, f & 1, f >> 1);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (q == NB2 - 1) VS_T(2);
                if (q == Q2 - 1) VS_T(3);
            }
        }
    };
    if constexpr (OVERLAP) {
        // the first tile (see the prologue), then the loop -- entered from that tile only, so that the compiler knows every
        // Yhat fragment to have arrived and puts no vmcnt wait into the loop (one there would also wait for the LDS-DMA
        // requests, which it cannot see)
        if (t_begin < t_end) {
            tile_body(t_begin, std::false_type{});
            for (int t = t_begin + 1; t < t_end; ++t) tile_body(t, std::false_type{});
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    } else {
        for (int t = t_begin; t < t_end; ++t) {
            if (MODE != VS_ACCUM && t * VS_BN + VS_BN > N) tile_body(t, std::true_type{});
            else tile_body(t, std::false_type{});
        }
    }
#ifdef VS_STAMP
    unsigned long long tloop;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tloop) :: "memory");
#endif

    const int m_out = m_row0;
    if constexpr (MODE == VS_ACCUM) {
        // U^T[c][m] sits with m on the lane; the consumer wants rows of U.  Transpose one 32x32 block at a time through
        // this wave's private 4.5 KB of LDS (the W stages are dead after the loop's last barrier) and store 128-byte
        // row segments: UT[split][m][c].
        constexpr int TP = 36;                                  // floats per transposed row (32 + 4 pad)
        __syncthreads();                                        // (slower waves may still be reading the last tile)
        float* tw = (float*)smem + wave * 32 * TP;
        const int64_t o_row0 = ((int64_t)split * m_pad + bx * VS_BM + wave * 32) * Cfg::KP;
        float* o = ut + o_row0;
        bf16_raw* ob = (bf16_raw*)ut + o_row0;               // bf16 slabs (ut_bf16): half the bytes out here and into the finish step
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(float4*)(tw + l31 * TP + 8 * g4 + 4 * h) = make_float4(U[cb][4 * g4], U[cb][4 * g4 + 1], U[cb][4 * g4 + 2], U[cb][4 * g4 + 3]);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const float4 v = *(const float4*)(tw + row * TP + ch * 4);
                if (ut_bf16) *(uint2*)(ob + (int64_t)row * Cfg::KP + cb * 32 + ch * 4) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                else *(float4*)(o + (int64_t)row * Cfg::KP + cb * 32 + ch * 4) = v;
            }
        }
    } else if constexpr (MODE == VS_SUMEXP) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const float se = sumexp[mb] + __shfl_xor(sumexp[mb], 32, 64);
            if (h == 0 && m_out + 32 * mb < m_pad) ut[((int64_t)split * m_pad + m_out + 32 * mb) * Cfg::KP + ones_col] = se;
        }
    } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float bv = best[mb];
            int bi = best_idx[mb];
            const float ob = __shfl_xor(bv, 32, 64);
            const int oi = __shfl_xor(bi, 32, 64);
            if (ob > bv || (ob == bv && oi < bi)) { bv = ob; bi = oi; }
            if (h == 0 && m_out + 32 * mb < m_pad) {
                part_score[(int64_t)split * m_pad + m_out + 32 * mb] = bv;
                part_idx[(int64_t)split * m_pad + m_out + 32 * mb] = bi == 0x7fffffff ? 0x7fffffff : bi + n_base;
            }
        }
    }
#ifdef VS_STAMP
    if (tid == 0) {
        unsigned long long tend;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tend) :: "memory");
        const int wg = blockIdx.x + gridDim.x * blockIdx.y;
        if (wg < 4096) {
            for (int i = 0; i < 6; ++i) vs_dbg[wg * 8 + i] = seg[i];
            // (slot 5 carries the prologue -- entry to first tile -- times the tile count: the reader divides per tile)
            vs_dbg[wg * 8 + 5] = (tstart - t_entry) * (unsigned long long)(t_end - t_begin);
            vs_dbg[wg * 8 + 6] = tloop - tstart;
            vs_dbg[wg * 8 + 7] = tend - tloop;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------
// finish: per sample row  p_t, clip, xent, d loss / d y      (reference :1198-1203 + Keras batch mean)
// ------------------------------------------------------------------------------------------------------
// (UB: the `ut` slabs hold bf16 instead of fp32 -- ut_f reads element i of either)
template <bool UB> __device__ __forceinline__ float ut_f(const float* ut, int64_t i) {
    if constexpr (UB) return __uint_as_float((uint32_t)((const bf16_raw*)ut)[i] << 16);
    else return ut[i];
}
#define C1V_FINISH_PARAMS const float* __restrict__ ut, int splits, int kp, int m_pad, int ones_col, const T* __restrict__ yhat, \
    const T* __restrict__ w, const float* __restrict__ rnorm, float qscale, const int* __restrict__ ids, int B, int T_len, int V, \
    int B_global, float* __restrict__ xent_rows, c1v_seq dy, const float* __restrict__ y, int64_t ldy, const int* __restrict__ pos, \
    const int* __restrict__ ctl, const int* __restrict__ live_rows, const int* __restrict__ live_ids
#define C1V_FINISH_ARGS ut, splits, kp, m_pad, ones_col, yhat, w, rnorm, qscale, ids, B, T_len, V, B_global, xent_rows, dy, y, ldy, pos, ctl, \
    live_rows, live_ids
template <typename T, bool UB>
__device__ __forceinline__ void xent_finish_body(C1V_FINISH_PARAMS) {
    // one wave per sample row; everything is row-wise and coalesced.  The projection that backpropagates through the
    // l2-normalisation uses yhat = y * rnorm in fp32 when `y` is given: a bf16-rounded yhat leaves a residue ALONG y,
    // which the transposed final conv amplifies coherently (measured: 6x larger relative error downstream).
    const int lane = threadIdx.x & 63;
    const int M = B * T_len;
    int m = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int id, cr;
    if (live_rows) {
        // one wave per LIVE row (the grid still covers M rows: a captured launch cannot resize it; waves behind the last
        // live row leave at once): compact row -> sample row and target from the inverse table of c1v_live_rows
        cr = m;
        if (cr >= __builtin_amdgcn_readfirstlane(ctl[0])) return;
        m = __builtin_amdgcn_readfirstlane(live_rows[cr]);
        id = __builtin_amdgcn_readfirstlane(live_ids[cr]);
    } else {
        if (m >= M) return;
        id = ids[m];
        // live-row compaction: `cr` is the row of yhat / ut that belongs to sample row m; rows without a target have none
        // (wave-uniform values are moved to scalar registers so that the loops over the slabs stay scalar loops)
        cr = pos ? __builtin_amdgcn_readfirstlane(pos[m]) : m;
    }
    const int b = m / T_len, t = m - b * T_len;
    const bool mask = id >= 0;
    if (ctl) splits = __builtin_amdgcn_readfirstlane((cr >> 7) < ctl[4] ? 1 : ctl[2]);    // (the full row tiles have ONE slab)
    if (cr < 0) {                                   // wave-uniform: no loss term, zero gradient
        if (lane == 0) xent_rows[m] = 0.0f;
        if (dy.data != nullptr) {
            const int64_t drow0 = (int64_t)b * dy.bstride + (int64_t)(dy.off + t) * dy.cp;
            // (bf16 rows of whole 16-byte words: one store per lane; the channel padding behind V is zero anyway)
            if (dy.dtype == C1V_BF16 && ((((uintptr_t)dy.data) & 15) == 0) && (dy.cp & 7) == 0 && (dy.bstride & 7) == 0 && ((V + 7) & ~7) <= dy.cp) {
                for (int c = lane * 8; c < V; c += 512) *(uint4*)((bf16_raw*)dy.data + drow0 + c) = make_uint4(0u, 0u, 0u, 0u);
            } else {
                for (int c = lane; c < V; c += 64) store_any(dy.data, dy.dtype, drow0 + c, 0.0f);
            }
        }
        return;
    }
    const float inv_q = 1.0f / qscale;
    const float inv_tau_eff = qscale * 0.6931471805599453f;     // 1/tau = qscale * ln 2
    // Training row with a target, rows of at most 512 columns (the common case): every lane owns 4-column chunks
    // (columns 4 (lane + 64 q) ..) and moves them as 8- / 16-byte words.  The kernel is bound by the NUMBER of memory
    // instructions a CU can issue (measured: its time scales with 1/occupancy and one-column-per-lane loads need ~60
    // instructions per row); chunks need ~20, and the softmax denominator comes with the chunk that holds `ones_col`.
    constexpr int NCH = 2;
    if (mask && dy.data != nullptr && V <= 256 * NCH && (kp & 3) == 0 && kp <= 256 * NCH && ones_col >= V &&
        ((((uintptr_t)ut) & 15) == 0) && ((((uintptr_t)w) | ((uintptr_t)yhat)) & 15) == 0) {
        float us[NCH][4], wv[NCH][4], yh[NCH][4], yv[NCH][4];
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) { us[q][e] = 0.f; wv[q][e] = 0.f; yh[q][e] = 0.f; yv[q][e] = 0.f; }
        // Loads are UNCONDITIONAL (the chunk index is clamped into the row and the value masked afterwards): a load under
        // `if (column < width)` compiles to a branch with its own s_waitcnt, i.e. one full memory round trip per load
        // (that is what the one-column-per-lane path below does: ~60 dependent round trips per row).
        int cc[NCH];
        bool in_kp[NCH];
#pragma unroll
        for (int q = 0; q < NCH; ++q) { const int c0 = 4 * (lane + 64 * q); in_kp[q] = c0 < kp; cc[q] = in_kp[q] ? c0 : kp - 4; }
        // (the first 64 steps of the sample's target count ride in the same batch of requests)
        const int tl = lane < T_len ? lane : T_len - 1;
        const int id_l = ids[b * T_len + tl];
        const float rn1 = rnorm[m];
        const bool y2 = y != nullptr && ((((uintptr_t)y) & 7) == 0) && (ldy & 1) == 0 && (V & 1) == 0 && V >= 4;
        // raw words only: nothing is converted (= waited for) until the slab requests below are out as well
        uint4 wraw[NCH], hraw[NCH];
        float2 ya[NCH], yb[NCH];
        float ye[NCH][4];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            wraw[q] = make_uint4(0u, 0u, 0u, 0u); hraw[q] = wraw[q];
            ya[q] = make_float2(0.f, 0.f); yb[q] = ya[q];
            if constexpr (sizeof(T) == 2) {
                const uint2 a = *(const uint2*)((const bf16_raw*)w + (int64_t)id * kp + cc[q]);
                const uint2 h2 = *(const uint2*)((const bf16_raw*)yhat + (int64_t)cr * kp + cc[q]);
                wraw[q].x = a.x; wraw[q].y = a.y; hraw[q].x = h2.x; hraw[q].y = h2.y;
            } else {
                wraw[q] = *(const uint4*)((const float*)w + (int64_t)id * kp + cc[q]);
                hraw[q] = *(const uint4*)((const float*)yhat + (int64_t)cr * kp + cc[q]);
            }
            if (y2) {
                // the chunk may hang over the end of the (unpadded) row: read the last whole chunk instead and shift
                const int c0 = 4 * (lane + 64 * q);
                const int cy = c0 + 3 < V ? c0 : V - 4;
                const float* yr = y + (int64_t)m * ldy + cy;
                ya[q] = *(const float2*)yr; yb[q] = *(const float2*)(yr + 2);
            } else if (y != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * (lane + 64 * q) + e;
                    ye[q][e] = y[(int64_t)m * ldy + (c < V ? c : V - 1)];
                }
            }
        }
        // partial sums of the vocabulary splits, four slabs (8 requests) per round, in slab order
        for (int sp0 = 0; sp0 < splits; sp0 += 4) {
            float4 v[4][NCH];
            if constexpr (UB) {
                uint2 r[4][NCH];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sp = sp0 + i < splits ? sp0 + i : splits - 1;
                    const bf16_raw* up = (const bf16_raw*)ut + ((int64_t)sp * m_pad + cr) * kp;
#pragma unroll
                    for (int q = 0; q < NCH; ++q) r[i][q] = *(const uint2*)(up + cc[q]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < NCH; ++q)
                        v[i][q] = make_float4(__uint_as_float(r[i][q].x << 16), __uint_as_float(r[i][q].x & 0xFFFF0000u),
                                              __uint_as_float(r[i][q].y << 16), __uint_as_float(r[i][q].y & 0xFFFF0000u));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sp = sp0 + i < splits ? sp0 + i : splits - 1;
                    const float* up = ut + ((int64_t)sp * m_pad + cr) * kp;
#pragma unroll
                    for (int q = 0; q < NCH; ++q) v[i][q] = *(const float4*)(up + cc[q]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (sp0 + i < splits) {
#pragma unroll
                    for (int q = 0; q < NCH; ++q) { us[q][0] += v[i][q].x; us[q][1] += v[i][q].y; us[q][2] += v[i][q].z; us[q][3] += v[i][q].w; }
                }
        }
        float cnt1 = (lane < T_len && id_l >= 0) ? 1.f : 0.f;
        for (int tt = lane + 64; tt < T_len; tt += 64) cnt1 += ids[b * T_len + tt] >= 0 ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            if constexpr (sizeof(T) == 2) {
                wv[q][0] = __uint_as_float(wraw[q].x << 16); wv[q][1] = __uint_as_float(wraw[q].x & 0xFFFF0000u);
                wv[q][2] = __uint_as_float(wraw[q].y << 16); wv[q][3] = __uint_as_float(wraw[q].y & 0xFFFF0000u);
                yh[q][0] = __uint_as_float(hraw[q].x << 16); yh[q][1] = __uint_as_float(hraw[q].x & 0xFFFF0000u);
                yh[q][2] = __uint_as_float(hraw[q].y << 16); yh[q][3] = __uint_as_float(hraw[q].y & 0xFFFF0000u);
            } else {
                wv[q][0] = __uint_as_float(wraw[q].x); wv[q][1] = __uint_as_float(wraw[q].y);
                wv[q][2] = __uint_as_float(wraw[q].z); wv[q][3] = __uint_as_float(wraw[q].w);
                yh[q][0] = __uint_as_float(hraw[q].x); yh[q][1] = __uint_as_float(hraw[q].y);
                yh[q][2] = __uint_as_float(hraw[q].z); yh[q][3] = __uint_as_float(hraw[q].w);
            }
            if (y2) {
                const int c0 = 4 * (lane + 64 * q);
                const int sh = c0 - (c0 + 3 < V ? c0 : V - 4);   // 0, or 2 for the chunk that straddles the end (V even)
                const float t4[4] = {ya[q].x, ya[q].y, yb[q].x, yb[q].y};
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[q][e] = (e + sh < 4) ? (sh == 0 ? t4[e] : t4[(e + 2) & 3]) : 0.f;
            } else if (y != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[q][e] = ye[q][e];
            }
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q)
            if (!in_kp[q]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { us[q][e] = 0.f; wv[q][e] = 0.f; yh[q][e] = 0.f; }
            }
        // softmax denominator: element `ones_col` of the summed row (w carries 1.0 in that column)
        const int ocj = ones_col >> 2, oce = ones_col & 3;
        float se = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((ocj >> 6) == q && oce == e) se = __shfl(us[q][e], ocj & 63, 64);
        float st1 = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * (lane + 64 * q) + e < V) st1 += yh[q][e] * wv[q][e];
        cnt1 = wave_sum_dpp(cnt1);
        st1 = wave_sum_dpp(st1);
        const float p1 = exp2f(st1) / se;
        const bool live1 = p1 > 1e-7f && p1 < (1.0f - 1e-7f);
        if (lane == 0) xent_rows[m] = -logf(fminf(fmaxf(p1, 1e-7f), 1.0f - 1e-7f));
        const int64_t drow1 = (int64_t)b * dy.bstride + (int64_t)(dy.off + t) * dy.cp;
        const float coef1 = live1 ? inv_tau_eff / (cnt1 * (float)B_global) : 0.0f;     // clipped: zero gradient
        const float inv_se1 = 1.0f / se;
        const bool normal1 = rn1 < 0.99e6f;                      // sum y^2 >= 1e-12
        float dot1 = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = 4 * (lane + 64 * q) + e < V;
                us[q][e] = in ? coef1 * (us[q][e] * inv_se1 - wv[q][e]) : 0.f;
                yv[q][e] = in ? (y ? yv[q][e] * rn1 : yh[q][e] * inv_q) : 0.f;
                dot1 += us[q][e] * yv[q][e];
            }
        dot1 = wave_sum_dpp(dot1);
        const bool st8 = dy.dtype == C1V_BF16 && ((((uintptr_t)dy.data) & 7) == 0) && (dy.cp & 3) == 0 && (dy.bstride & 3) == 0;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c0 = 4 * (lane + 64 * q);
            if (c0 >= V) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = coef1 == 0.0f ? 0.0f : (normal1 ? (us[q][e] - yv[q][e] * dot1) * rn1 : us[q][e] * rn1);
            if (st8 && c0 + 3 < V) {
                *(uint2*)((bf16_raw*)dy.data + drow1 + c0) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c0 + e < V) store_any(dy.data, dy.dtype, drow1 + c0 + e, o[e]);
            }
        }
        return;
    }
    float sumexp = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < splits; ++sp) sumexp += ut_f<UB>(ut, ((int64_t)sp * m_pad + cr) * kp + ones_col);
    float cnt = 0.f;
    for (int tt = lane; tt < T_len; tt += 64) cnt += ids[b * T_len + tt] >= 0 ? 1.f : 0.f;
    float st = 0.f;
    if (mask)
        for (int c = lane; c < V; c += 64)
            st += ElemTraits<T>::load(yhat, (int64_t)cr * kp + c) * ElemTraits<T>::load(w, (int64_t)id * kp + c);
    cnt = wave_sum_dpp(cnt);
    st = wave_sum_dpp(st);
    const float p_t = exp2f(st) / sumexp;
    const bool live = mask && p_t > 1e-7f && p_t < (1.0f - 1e-7f);
    const float p_c = fminf(fmaxf(p_t, 1e-7f), 1.0f - 1e-7f);
    if (lane == 0) xent_rows[m] = mask ? -logf(p_c) : 0.0f;
    if (dy.data == nullptr) return;
    const float coef = live ? inv_tau_eff / (cnt * (float)B_global) : 0.0f;
    const float inv_se = 1.0f / sumexp;
    const int64_t drow = (int64_t)b * dy.bstride + (int64_t)(dy.off + t) * dy.cp;
    if (coef == 0.0f) {                                          // masked / clipped row: zero gradient (wave-uniform branch)
        for (int c = lane; c < V; c += 64) store_any(dy.data, dy.dtype, drow + c, 0.0f);
        return;
    }
    // u (summed over splits), yhat and the target row stay in registers between the two passes (V <= 64*UC)
    constexpr int UC = 8;
    float uu[UC], yy[UC];
    float dot = 0.f;
    const float rn = rnorm[m];
    const bool normal = rn < 0.99e6f;                           // sum y^2 >= 1e-12
    if (V <= 64 * UC) {
        // slab loop outside, column loop (unrolled) inside: UC independent loads per slab are in flight together -- with the
        // loops the other way round every load waited for the one before it (splits x UC serial round trips per row)
#pragma unroll
        for (int k = 0; k < UC; ++k) uu[k] = 0.f;
#pragma unroll 2
        for (int sp = 0; sp < splits; ++sp) {
            const int64_t up = ((int64_t)sp * m_pad + cr) * kp;
#pragma unroll
            for (int k = 0; k < UC; ++k) {
                const int c = lane + 64 * k;
                if (c < V) uu[k] += ut_f<UB>(ut, up + c);
            }
        }
#pragma unroll
        for (int k = 0; k < UC; ++k) {
            const int c = lane + 64 * k;
            yy[k] = 0.f;
            if (c < V) {
                const float u = uu[k];
                uu[k] = coef * (u * inv_se - ElemTraits<T>::load(w, (int64_t)id * kp + c));
                yy[k] = y ? y[(int64_t)m * ldy + c] * rn : ElemTraits<T>::load(yhat, (int64_t)cr * kp + c) * inv_q;
                dot += uu[k] * yy[k];
            }
        }
        dot = wave_sum_dpp(dot);
#pragma unroll
        for (int k = 0; k < UC; ++k) {
            const int c = lane + 64 * k;
            if (c < V) store_any(dy.data, dy.dtype, drow + c, normal ? (uu[k] - yy[k] * dot) * rn : uu[k] * rn);
        }
        return;
    }
    for (int c = lane; c < V; c += 64) {
        float u = 0.f;
        for (int sp = 0; sp < splits; ++sp) u += ut_f<UB>(ut, ((int64_t)sp * m_pad + cr) * kp + c);
        const float dyh = coef * (u * inv_se - ElemTraits<T>::load(w, (int64_t)id * kp + c));
        dot += dyh * (y ? y[(int64_t)m * ldy + c] * rn : ElemTraits<T>::load(yhat, (int64_t)cr * kp + c) * inv_q);
    }
    dot = wave_sum_dpp(dot);
    for (int c = lane; c < V; c += 64) {
        float u = 0.f;
        for (int sp = 0; sp < splits; ++sp) u += ut_f<UB>(ut, ((int64_t)sp * m_pad + cr) * kp + c);
        const float dyh = coef * (u * inv_se - ElemTraits<T>::load(w, (int64_t)id * kp + c));
        const float yh = y ? y[(int64_t)m * ldy + c] * rn : ElemTraits<T>::load(yhat, (int64_t)cr * kp + c) * inv_q;
        store_any(dy.data, dy.dtype, drow + c, normal ? (dyh - yh * dot) * rn : dyh * rn);
    }
}
template <typename T, bool UB>
__global__ __launch_bounds__(256) void xent_finish_kernel(C1V_FINISH_PARAMS) { xent_finish_body<T, UB>(C1V_FINISH_ARGS); }
// The training instance (bf16 operands, bf16 slabs) at EIGHT waves per SIMD: the kernel is one wave per row and two dependent
// memory round trips long, so its time is rows / resident waves.  The compiler wanted 66 registers (7 waves); held to 64 it
// spills three dwords -- C2's 8192 rows then fit ONE round of the chip's 8192 wave slots instead of 1.14.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void xent_finish_bf16_kernel(const float* __restrict__ ut, int splits, int kp, int m_pad, int ones_col, const bf16_raw* __restrict__ yhat,
                             const bf16_raw* __restrict__ w, const float* __restrict__ rnorm, float qscale, const int* __restrict__ ids, int B,
                             int T_len, int V, int B_global, float* __restrict__ xent_rows, c1v_seq dy, const float* __restrict__ y, int64_t ldy,
                             const int* __restrict__ pos, const int* __restrict__ ctl, const int* __restrict__ live_rows,
                             const int* __restrict__ live_ids) {
    xent_finish_body<bf16_raw, true>(C1V_FINISH_ARGS);
}

// loss scalars: see c1v_lossred.h
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ xent_rows, const int* __restrict__ ids,
                                                           const float* __restrict__ kl_b, int B, int T_len, int B_global,
                                                           float* __restrict__ out3, float* __restrict__ xent_b) {
    loss_reduce_body(xent_rows, ids, kl_b, B, T_len, B_global, out3, xent_b);
}

// ------------------------------------------------------------------------------------------------------
// arg-max helpers
// ------------------------------------------------------------------------------------------------------
// merge R partial (score, idx) rows: highest score, ties -> lowest index
__global__ void argmax_merge_kernel(const float* __restrict__ sc, const int* __restrict__ ix, int R, int64_t stride, int M,
                                    float* __restrict__ score, int* __restrict__ idx) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float best = sc[m]; int bi = ix[m];
    for (int r = 1; r < R; ++r) {
        const float v = sc[(int64_t)r * stride + m]; const int i = ix[(int64_t)r * stride + m];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    score[m] = best; idx[m] = bi;
}

// fp32 path: arg-max over a chunk of materialised logits [M][ld], merged into running (score, idx)
__global__ __launch_bounds__(256) void row_argmax_kernel(const float* __restrict__ logits, int M, int64_t ld, int n_count, int n_first,
                                                         float* __restrict__ score, int* __restrict__ idx, int first_chunk) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float best = -3.0e38f; int bi = 0x7fffffff;
    for (int n = lane; n < n_count; n += 64) {
        const float v = logits[(int64_t)row * ld + n];
        if (v > best) { best = v; bi = n + n_first; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        if (!first_chunk) { const float pv = score[row]; const int pi = idx[row]; if (pv > best || (pv == best && pi < bi)) { best = pv; bi = pi; } }
        score[row] = best; idx[row] = bi;
    }
}

// top-n of every row of materialised logits [M][ld] (a10-ii with topn > 1: gensim `similar_by_vector(v, topn)`).
// Round r selects the best entry strictly after round r-1's winner in the total order (score descending, index
// ascending), so no per-lane candidate lists are needed; n rounds re-read the row from L2/HBM (topn is small).
__global__ __launch_bounds__(256) void row_topn_kernel(const float* __restrict__ logits, int M, int64_t ld, int N, int n_base, int topn,
                                                       float* __restrict__ score, int* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* lr = logits + (int64_t)row * ld;
    float prev_v = 3.4e38f; int prev_i = -1;
    for (int r = 0; r < topn; ++r) {
        float best = -3.4e38f; int bi = 0x7fffffff;
        for (int n = lane; n < N; n += 64) {
            const float v = lr[n];
            const bool after = (v < prev_v) || (v == prev_v && n > prev_i);
            if (after && v > best) { best = v; bi = n; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { score[(int64_t)row * topn + r] = best; idx[(int64_t)row * topn + r] = bi == 0x7fffffff ? -1 : bi + n_base; }
        prev_v = best; prev_i = bi;
    }
}

// ------------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------------
extern "C" int c1v_l2norm_rows(const c1v_l2norm_args* a, void* stream) {
    C1V_REQUIRE(a && a->y && a->yhat, "l2norm: null argument");
    C1V_REQUIRE(a->M > 0 && a->V > 0 && a->kp >= a->V, "l2norm: bad shape M=%d V=%d kp=%d", a->M, a->V, a->kp);
    dim3 grid((unsigned)c1v_cdiv(a->M, 4));
    if (a->dtype == C1V_F32)
        hipLaunchKernelGGL(l2norm_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a->y, a->M, a->V, a->ldy, (float*)a->yhat, a->kp, a->rnorm, a->qscale, a->pos);
    else
        hipLaunchKernelGGL(l2norm_rows_kernel<bf16_raw>, grid, dim3(256), 0, (hipStream_t)stream, a->y, a->M, a->V, a->ldy, (bf16_raw*)a->yhat, a->kp, a->rnorm, a->qscale, a->pos);
    C1V_LAUNCH_CHECK();
    return 0;
}

template <int CB, int MODE, int MB>
static int launch_vs_mb(const bf16_raw* yhat, int M, const bf16_raw* w, int N, int n_base, int splits, float* ut, int ut_bf16, int m_pad, int ones_col,
                        float* ps, int* pi, const int* ctl, hipStream_t st) {
    using Cfg = VsCfg<CB>;
    static bool once = (hipFuncSetAttribute((const void*)vocab_stream_kernel<CB, MODE, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, VS_STAGES * Cfg::STAGE), true);
    (void)once;
    const int tiles_total = (int)c1v_cdiv(N, VS_BN);
    const int tps = (int)c1v_cdiv(tiles_total, splits);
    dim3 grid((unsigned)c1v_cdiv(m_pad, VS_BM * MB), (unsigned)splits);
    hipLaunchKernelGGL((vocab_stream_kernel<CB, MODE, MB>), grid, dim3(256), VS_STAGES * Cfg::STAGE, st, yhat, M, w, N, n_base, tiles_total, tps, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl);
    C1V_LAUNCH_CHECK();
    return 0;
}

template <int CB, int MODE>
static int launch_vs(const bf16_raw* yhat, int M, const bf16_raw* w, int N, int n_base, int splits, float* ut, int ut_bf16, int m_pad, int ones_col,
                     float* ps, int* pi, const int* ctl, hipStream_t st) {
    if constexpr (MODE != VS_ACCUM) {
        // 256-row workgroups when they still fill the chip (and the grid is not a device-dealt pool)
        if (ctl == nullptr && c1v_cdiv(m_pad, 2 * VS_BM) * splits >= 256)
            return launch_vs_mb<CB, MODE, 2>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
    }
    return launch_vs_mb<CB, MODE, 1>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
}

// widths (elements per row of Yhat / W) the fused stream kernel is instantiated for
static inline bool vs_fused_width(int kp) { return kp == 32 || (kp % 64 == 0 && kp >= 64 && kp <= 320); }

template <int MODE>
static int dispatch_vs(int kp, const bf16_raw* yhat, int M, const bf16_raw* w, int N, int n_base, int splits, float* ut, int ut_bf16, int m_pad, int ones_col,
                       float* ps, int* pi, const int* ctl, hipStream_t st) {
    switch (kp) {
        case 32:  return launch_vs<1, MODE>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
        case 64:  return launch_vs<2, MODE>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
        case 128: return launch_vs<4, MODE>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
        case 192: return launch_vs<6, MODE>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
        case 256: return launch_vs<8, MODE>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
        case 320: return launch_vs<10, MODE>(yhat, M, w, N, n_base, splits, ut, ut_bf16, m_pad, ones_col, ps, pi, ctl, st);
        default: return 1;   // no fused instantiation for this width
    }
}

__global__ void zero_f32_kernel(float* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

// unfused path (fp32, and bf16 widths without a fused instantiation): NT GEMM (+exp2) into `logits`, TN GEMM out.
static int xent_accum_unfused(const c1v_xent_accum_args* a, hipStream_t st) {
    C1V_REQUIRE(a->logits && a->n_chunk > 0, "cos_softmax_accum: this dtype/width needs the `logits` scratch");
    const int es = c1v_esize(a->dtype);
    const int chunks = (int)c1v_cdiv(a->N, a->n_chunk);
    C1V_REQUIRE(a->splits % chunks == 0, "cos_softmax_accum: splits (%d) must be a multiple of the chunk count (%d)", a->splits, chunks);
    const int spc = a->splits / chunks;
    for (int ch = 0; ch < chunks; ++ch) {
        const int nc0 = ch * a->n_chunk;
        const int ncount = (a->N - nc0) < a->n_chunk ? (a->N - nc0) : a->n_chunk;
        // E[n][m] = exp2(sum_k W[nc0+n][k] * yhat[m][k])     stored in the compute dtype [n_chunk][m_pad]
        NtParams p{};
        p.A = (const char*)a->w + (int64_t)nc0 * a->kp * es; p.a_bstride_b = (int64_t)a->kp * es; p.lda_b = (int64_t)a->kp * es; p.a_off = 0;
        p.W = (const char*)a->yhat; p.ldw_b = (int64_t)a->kp * es; p.Kb = (int64_t)a->kp * es;
        EpiDev& e = p.e;
        e = EpiDev{};
        e.exp2 = 1; e.row_mode = C1V_ROW_STORE;
        e.dst = a->logits; e.dst_dtype = a->dtype; e.dst_bstride = a->m_pad; e.dst_off = 0; e.dst_len = 1; e.dst_cp = a->m_pad; e.dst_c = a->M;
        e.act_rowmul = 1; e.M = ncount; e.Le = 1; e.nrows = 1; e.N = a->M;
        int rc = c1v_launch_nt(a->dtype, p, st);
        if (rc) return rc;
        if (chunks > 1 && ncount < a->n_chunk) {   // rows of the tile that straddles the end still hold the previous chunk
            const int64_t tail = c1v_cdiv(ncount, 64) * 64 - ncount;
            if (tail > 0) (void)hipMemsetAsync((char*)a->logits + (int64_t)ncount * a->m_pad * es, 0, (size_t)tail * a->m_pad * es, st);
        }
        if (!a->need_u) {
            // only the denominator: column sums of E over n  == row `ones_col` of U^T
            TnParams t{};
            t.A = (const char*)a->w + (int64_t)nc0 * a->kp * es + (int64_t)(a->ones_col / 128 * 128) * es; t.lda_b = (int64_t)a->kp * es;
            t.I = a->ones_col % 128 + 1; t.Ia = (a->kp - a->ones_col / 128 * 128) < 128 ? (a->kp - a->ones_col / 128 * 128) : 128;
            t.D = (const char*)a->logits; t.ldd_b = (int64_t)a->m_pad * es; t.J = a->M; t.Ja = a->m_pad;
            t.r_begin = 0; t.r_end = ncount; t.rows_per_split = c1v_cdiv(c1v_cdiv(ncount, spc), 64) * 64;   // whole tiles per split
            t.a_shift0 = 0; t.taps = 1; t.flip = 0; t.tap_stride = 0; t.si = 1; t.sj = a->kp;
            t.out = (float*)a->ut + (int64_t)ch * spc * a->kp * a->m_pad + (a->ones_col / 128 * 128); t.slab_stride = (int64_t)a->kp * a->m_pad; t.splits = spc;
            rc = c1v_launch_tn(a->dtype, t, st);
        } else {
            TnParams t{};
            t.A = (const char*)a->w + (int64_t)nc0 * a->kp * es; t.lda_b = (int64_t)a->kp * es; t.I = a->kp; t.Ia = a->kp;
            t.D = (const char*)a->logits; t.ldd_b = (int64_t)a->m_pad * es; t.J = a->M; t.Ja = a->m_pad;
            t.r_begin = 0; t.r_end = ncount; t.rows_per_split = c1v_cdiv(c1v_cdiv(ncount, spc), 64) * 64;   // whole tiles per split
            t.a_shift0 = 0; t.taps = 1; t.flip = 0; t.tap_stride = 0; t.si = 1; t.sj = a->kp;
            t.out = (float*)a->ut + (int64_t)ch * spc * a->kp * a->m_pad; t.slab_stride = (int64_t)a->kp * a->m_pad; t.splits = spc;
            rc = c1v_launch_tn(a->dtype, t, st);
        }
        if (rc) return rc;
    }
    return 0;
}

extern "C" int c1v_cos_softmax_accum(const c1v_xent_accum_args* a, void* stream) {
    C1V_REQUIRE(a && a->yhat && a->w && a->ut, "cos_softmax_accum: null argument");
    C1V_REQUIRE(a->M > 0 && a->N > 0 && a->kp % 32 == 0 && a->splits > 0, "cos_softmax_accum: bad shape M=%d N=%d kp=%d splits=%d", a->M, a->N, a->kp, a->splits);
    C1V_REQUIRE(a->m_pad % VS_BM == 0 && a->m_pad >= a->M, "cos_softmax_accum: m_pad (%d) must be a multiple of %d covering M", a->m_pad, VS_BM);
    C1V_REQUIRE(a->n_pad % VS_BN == 0 && a->n_pad >= a->N, "cos_softmax_accum: n_pad (%d) must be a multiple of %d covering N", a->n_pad, VS_BN);
    C1V_REQUIRE(a->ones_col >= 0 && a->ones_col < a->kp, "cos_softmax_accum: ones_col out of range");
    hipStream_t st = (hipStream_t)stream;
    const int ub = a->ut_dtype == C1V_BF16;
    C1V_REQUIRE(a->ut_dtype == C1V_F32 || (ub && a->dtype == C1V_BF16 && a->need_u && vs_fused_width(a->kp)),
                "cos_softmax_accum: bf16 `ut` slabs are written by the fused bf16 kernel with need_u = 1 only (dtype %d, need_u %d, kp %d)",
                a->dtype, a->need_u, a->kp);
    if (a->dtype == C1V_BF16) {
        if (a->live_ctl) {
            C1V_REQUIRE(vs_fused_width(a->kp), "cos_softmax_accum: live_ctl needs the fused stream kernel (kp = 32 or a multiple of 64 up to 320; got %d)", a->kp);
        }
        int rc = a->need_u
            ? dispatch_vs<VS_ACCUM>(a->kp, (const bf16_raw*)a->yhat, a->M, (const bf16_raw*)a->w, a->N, 0, a->splits, (float*)a->ut, ub, a->m_pad, a->ones_col, nullptr, nullptr, a->live_ctl, st)
            : dispatch_vs<VS_SUMEXP>(a->kp, (const bf16_raw*)a->yhat, a->M, (const bf16_raw*)a->w, a->N, 0, a->splits, (float*)a->ut, 0, a->m_pad, a->ones_col, nullptr, nullptr, a->live_ctl, st);
        if (rc <= 0) return rc;
    }
    C1V_REQUIRE(a->live_ctl == nullptr, "cos_softmax_accum: live_ctl is only supported by the fused bf16 stream kernel");
    return xent_accum_unfused(a, st);
}

extern "C" int c1v_live_rows(const int32_t* ids, int32_t M, int32_t items, int32_t vocab_tiles, int32_t max_splits, int32_t* pos, int32_t* ctl,
                             void* stream) {
    C1V_REQUIRE(ids && pos && ctl && M > 0 && items > 0 && vocab_tiles > 0 && max_splits > 0, "live_rows: bad argument");
    C1V_REQUIRE(items >= (int)c1v_cdiv(M, VS_BM), "live_rows: the stream grid (%d workgroups) must cover all %d row tiles", items, (int)c1v_cdiv(M, VS_BM));
    hipLaunchKernelGGL(live_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, M, items, vocab_tiles, max_splits, pos, ctl,
                       (int*)nullptr, (int*)nullptr);
    C1V_LAUNCH_CHECK();
    return 0;
}
extern "C" int c1v_live_rows_table(const c1v_live_rows_args* a, void* stream) {
    C1V_REQUIRE(a && a->ids && a->pos && a->ctl && a->M > 0 && a->items > 0 && a->vocab_tiles > 0 && a->max_splits > 0, "live_rows_table: bad argument");
    C1V_REQUIRE(a->items >= (int)c1v_cdiv(a->M, VS_BM), "live_rows_table: the stream grid (%d workgroups) must cover all %d row tiles", a->items, (int)c1v_cdiv(a->M, VS_BM));
    C1V_REQUIRE((a->rows != nullptr) == (a->cids != nullptr), "live_rows_table: rows / cids come together");
    hipLaunchKernelGGL(live_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a->ids, a->M, a->items, a->vocab_tiles, a->max_splits,
                       a->pos, a->ctl, a->rows, a->cids);
    C1V_LAUNCH_CHECK();
    return 0;
}
extern "C" int c1v_cos_softmax_xent_fwd(const c1v_xent_accum_args* a, void* stream) {
    C1V_REQUIRE(a, "null argument");
    c1v_xent_accum_args b = *a; b.need_u = 0; return c1v_cos_softmax_accum(&b, stream);
}
extern "C" int c1v_cos_softmax_xent_bwd(const c1v_xent_accum_args* a, void* stream) {
    C1V_REQUIRE(a, "null argument");
    c1v_xent_accum_args b = *a; b.need_u = 1; return c1v_cos_softmax_accum(&b, stream);
}

extern "C" int c1v_cos_softmax_finish(const c1v_xent_finish_args* a, void* stream) {
    C1V_REQUIRE(a && a->ut && a->yhat && a->w && a->ids && a->xent_rows && a->rnorm, "cos_softmax_finish: null argument");
    C1V_REQUIRE(a->B > 0 && a->T > 0 && a->V > 0 && a->V <= a->kp && a->splits > 0, "cos_softmax_finish: bad shape");
    C1V_REQUIRE(a->m_pad >= (int)(c1v_cdiv((int64_t)a->B * a->T, 32) * 32), "cos_softmax_finish: m_pad too small");
    C1V_REQUIRE(a->dy.data == nullptr || (a->dy.cp >= a->V && a->dy.len >= a->T), "cos_softmax_finish: dy buffer too small");
    const int M = a->B * a->T;
    static const size_t lds = getenv("C1V_FINISH_LDS") ? (size_t)atoi(getenv("C1V_FINISH_LDS")) : 0;   // diagnostic: caps the resident waves per CU
    dim3 grid((unsigned)c1v_cdiv(M, 4));
    C1V_REQUIRE(a->ut_dtype == C1V_F32 || (a->ut_dtype == C1V_BF16 && a->dtype == C1V_BF16), "cos_softmax_finish: bf16 `ut` slabs come from the bf16 kernel only");
    C1V_REQUIRE((a->live_rows != nullptr) == (a->live_ids != nullptr) && (a->live_rows == nullptr || (a->pos && a->live_ctl)),
                "cos_softmax_finish: live_rows / live_ids come together, with pos and live_ctl");
#define C1V_FINISH(ELEM, UB) hipLaunchKernelGGL((xent_finish_kernel<ELEM, UB>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)a->ut, a->splits, a->kp, \
        a->m_pad, a->ones_col, (const ELEM*)a->yhat, (const ELEM*)a->w, a->rnorm, a->qscale, a->ids, a->B, a->T, a->V, a->B_global, a->xent_rows, a->dy, a->y, \
        a->ldy, a->pos, a->live_ctl, a->live_rows, a->live_ids)
    if (a->dtype == C1V_F32) C1V_FINISH(float, false);
    else if (a->ut_dtype == C1V_BF16)
        hipLaunchKernelGGL(xent_finish_bf16_kernel, grid, dim3(256), lds, (hipStream_t)stream, (const float*)a->ut, a->splits, a->kp, a->m_pad,
                           a->ones_col, (const bf16_raw*)a->yhat, (const bf16_raw*)a->w, a->rnorm, a->qscale, a->ids, a->B, a->T, a->V, a->B_global,
                           a->xent_rows, a->dy, a->y, a->ldy, a->pos, a->live_ctl, a->live_rows, a->live_ids);
    else C1V_FINISH(bf16_raw, false);
#undef C1V_FINISH
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_loss_reduce(const float* xent_rows, const int32_t* ids, const float* kl_b, int32_t B, int32_t T, int32_t B_global,
                               float* out3, float* xent_b, void* stream) {
    C1V_REQUIRE(xent_rows && ids && out3 && B > 0 && T > 0 && B_global > 0, "loss_reduce: bad argument");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, xent_rows, ids, kl_b, B, T, B_global, out3, xent_b);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_argmax_merge(const float* scores, const int32_t* idxs, int32_t R, int32_t M, float* score, int32_t* idx, void* stream) {
    C1V_REQUIRE(scores && idxs && score && idx && R > 0 && M > 0, "argmax_merge: bad argument");
    hipLaunchKernelGGL(argmax_merge_kernel, dim3((unsigned)c1v_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, scores, idxs, R, (int64_t)M, M, score, idx);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_vocab_argmax(const c1v_vocab_argmax_args* a, void* stream) {
    C1V_REQUIRE(a && a->q && a->w && a->score && a->idx, "vocab_argmax: null argument");
    C1V_REQUIRE(a->M > 0 && a->N > 0 && a->kp % 32 == 0, "vocab_argmax: bad shape M=%d N=%d kp=%d", a->M, a->N, a->kp);
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == C1V_BF16 && a->part_score && a->part_idx && a->splits > 0 && a->m_pad % VS_BM == 0 && a->m_pad >= a->M &&
        a->n_pad % VS_BN == 0 && a->n_pad >= a->N) {
        int rc = dispatch_vs<VS_ARGMAX>(a->kp, (const bf16_raw*)a->q, a->M, (const bf16_raw*)a->w, a->N, a->n_base, a->splits, nullptr, 0, a->m_pad, 0,
                                        a->part_score, a->part_idx, nullptr, st);
        if (rc < 0) return rc;
        if (rc == 0) {
            hipLaunchKernelGGL(argmax_merge_kernel, dim3((unsigned)c1v_cdiv(a->M, 256)), dim3(256), 0, st, a->part_score, a->part_idx, a->splits,
                               (int64_t)a->m_pad, a->M, a->score, a->idx);
            C1V_LAUNCH_CHECK();
            return 0;
        }
    }
    // unfused: logits[m][n] per chunk of vocabulary rows, then row arg-max
    C1V_REQUIRE(a->logits && a->n_chunk > 0, "vocab_argmax: this dtype/width needs the `logits` scratch");
    const int es = c1v_esize(a->dtype);
    const int chunks = (int)c1v_cdiv(a->N, a->n_chunk);
    for (int ch = 0; ch < chunks; ++ch) {
        const int nc0 = ch * a->n_chunk;
        const int ncount = (a->N - nc0) < a->n_chunk ? (a->N - nc0) : a->n_chunk;
        NtParams p{};
        p.A = (const char*)a->q; p.a_bstride_b = (int64_t)a->kp * es; p.lda_b = (int64_t)a->kp * es; p.a_off = 0;
        p.W = (const char*)a->w + (int64_t)nc0 * a->kp * es; p.ldw_b = (int64_t)a->kp * es; p.Kb = (int64_t)a->kp * es;
        p.e = EpiDev{};
        p.e.row_mode = C1V_ROW_STORE; p.e.dst = a->logits; p.e.dst_dtype = C1V_F32; p.e.dst_bstride = a->n_chunk; p.e.dst_off = 0; p.e.dst_len = 1;
        p.e.dst_cp = a->n_chunk; p.e.dst_c = ncount; p.e.act_rowmul = 1; p.e.M = a->M; p.e.Le = 1; p.e.nrows = 1; p.e.N = ncount;
        int rc = c1v_launch_nt(a->dtype, p, st);
        if (rc) return rc;
        hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)c1v_cdiv(a->M, 4)), dim3(256), 0, st, a->logits, a->M, (int64_t)a->n_chunk, ncount,
                           nc0 + a->n_base, a->score, a->idx, ch == 0 ? 1 : 0);
        C1V_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int c1v_vocab_topn(const c1v_vocab_topn_args* a, void* stream) {
    C1V_REQUIRE(a && a->q && a->w && a->score && a->idx && a->logits, "vocab_topn: null argument");
    C1V_REQUIRE(a->dtype == C1V_F32 || a->dtype == C1V_BF16, "vocab_topn: bad dtype");
    C1V_REQUIRE(a->M > 0 && a->N > 0 && a->kp % 32 == 0 && a->ld >= a->N, "vocab_topn: bad shape M=%d N=%d kp=%d ld=%lld", a->M, a->N, a->kp, (long long)a->ld);
    C1V_REQUIRE(a->topn > 0 && a->topn <= a->N && a->topn <= 1024, "vocab_topn: topn (%d) must be in [1, min(N, 1024)]", a->topn);
    hipStream_t st = (hipStream_t)stream;
    const int es = c1v_esize(a->dtype);
    NtParams p{};
    p.A = (const char*)a->q; p.a_bstride_b = (int64_t)a->kp * es; p.lda_b = (int64_t)a->kp * es; p.a_off = 0;
    p.W = (const char*)a->w; p.ldw_b = (int64_t)a->kp * es; p.Kb = (int64_t)a->kp * es;
    p.e = EpiDev{};
    p.e.row_mode = C1V_ROW_STORE; p.e.dst = a->logits; p.e.dst_dtype = C1V_F32; p.e.dst_bstride = a->ld; p.e.dst_off = 0; p.e.dst_len = 1;
    p.e.dst_cp = (int)a->ld; p.e.dst_c = a->N; p.e.act_rowmul = 1; p.e.M = a->M; p.e.Le = 1; p.e.nrows = 1; p.e.N = a->N;
    C1V_REQUIRE(a->ld < (1ll << 31), "vocab_topn: ld too large");
    int rc = c1v_launch_nt(a->dtype, p, st);
    if (rc) return rc;
    hipLaunchKernelGGL(row_topn_kernel, dim3((unsigned)c1v_cdiv(a->M, 4)), dim3(256), 0, st, a->logits, a->M, a->ld, a->N, a->n_base, a->topn, a->score, a->idx);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// c1v_vocab_argmax for callers that cannot allocate: the per-split partial (score, index) rows live in a file-scope
// device array instead of caller scratch.  Every GPU holds its own copy of the array at its own address, looked up once
// per device with hipGetSymbolAddress; host code only ever uses that pointer (the array's NAME is a host-side stand-in,
// not a device address).
// ------------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char c1v_noalloc_scratch[C1V_NOALLOC_SCRATCH_BYTES];

#include <mutex>
static std::mutex g_noalloc_mu;
static void* g_noalloc_ptr[C1V_NOALLOC_MAX_DEVICES];          // device address of c1v_noalloc_scratch, per device

static int noalloc_scratch_ptr(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= C1V_NOALLOC_MAX_DEVICES) {
        c1v_set_error("vocab_argmax_noalloc: cannot identify the current device (%s, device %d)", hipGetErrorString(e), dev);
        return -1;
    }
    std::lock_guard<std::mutex> lock(g_noalloc_mu);
    if (g_noalloc_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(c1v_noalloc_scratch));      // with device `dev` current: ITS copy of the array
        if (e != hipSuccess || p == nullptr) {
            c1v_set_error("vocab_argmax_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return -1;
        }
        g_noalloc_ptr[dev] = p;
    }
    *out = g_noalloc_ptr[dev];
    return 0;
}

extern "C" int c1v_vocab_argmax_noalloc(const c1v_vocab_argmax_noalloc_args* a, void* stream) {
    C1V_REQUIRE(a && a->q && a->w && a->score && a->idx, "vocab_argmax_noalloc: null argument");
    C1V_REQUIRE(a->M > 0 && a->N > 0, "vocab_argmax_noalloc: bad shape M=%d N=%d", a->M, a->N);
    C1V_REQUIRE(vs_fused_width(a->kp),
                "vocab_argmax_noalloc: kp = %d has no fused bf16 kernel (32 or a multiple of 64 up to 320); use c1v_vocab_argmax with a logits scratch", a->kp);
    C1V_REQUIRE(a->n_pad % VS_BN == 0 && a->n_pad >= a->N, "vocab_argmax_noalloc: n_pad (%d) must be a multiple of %d covering N", a->n_pad, VS_BN);
    const int64_t m_pad = c1v_cdiv(a->M, VS_BM) * VS_BM;
    const int64_t per_split = m_pad * (int64_t)(sizeof(float) + sizeof(int32_t));
    C1V_REQUIRE(per_split <= (int64_t)C1V_NOALLOC_SCRATCH_BYTES,
                "vocab_argmax_noalloc: M = %d queries need %lld bytes of partial results, the static scratch holds %u (at most %d queries)",
                a->M, (long long)per_split, (unsigned)C1V_NOALLOC_SCRATCH_BYTES, C1V_NOALLOC_MAX_QUERIES);
    // vocabulary splits: enough workgroups to fill the chip (as the Python host picks them), capped by what the scratch holds
    int64_t splits = (2 * 256) / (m_pad / VS_BM);
    const int64_t tiles = a->n_pad / VS_BN;
    if (splits > tiles) splits = tiles;
    if (splits > 16) splits = 16;
    if (splits > (int64_t)C1V_NOALLOC_SCRATCH_BYTES / per_split) splits = (int64_t)C1V_NOALLOC_SCRATCH_BYTES / per_split;
    if (splits < 1) splits = 1;
    void* base = nullptr;
    if (noalloc_scratch_ptr(&base)) return -1;
    float* part_score = (float*)base;
    int32_t* part_idx = (int32_t*)((char*)base + splits * m_pad * (int64_t)sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    // every (split, row < m_pad) cell of the partial results is written by the stream kernel before the merge reads it, so
    // nothing of the previous call's contents has to be cleared
    int rc = dispatch_vs<VS_ARGMAX>(a->kp, (const bf16_raw*)a->q, a->M, (const bf16_raw*)a->w, a->N, a->n_base, (int)splits, nullptr, 0, (int)m_pad, 0,
                                    part_score, part_idx, nullptr, st);
    if (rc != 0) { if (rc > 0) c1v_set_error("vocab_argmax_noalloc: no fused kernel for kp = %d", a->kp); return rc < 0 ? rc : -1; }
    hipLaunchKernelGGL(argmax_merge_kernel, dim3((unsigned)c1v_cdiv(a->M, 256)), dim3(256), 0, st, part_score, part_idx, (int)splits, m_pad, a->M,
                       a->score, a->idx);
    C1V_LAUNCH_CHECK();
    return 0;
}
