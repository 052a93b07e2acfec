// One launch for the two GEMMs of a backward stage: the weight gradient (TN) and the data gradient (NT) both read the
// stage's dO buffer and neither fills the chip at this model's shapes (tens to a few hundred workgroups each), so they
// share one grid: workgroups [0, tn_blocks) run the wgrad tiles (the longer K loops, dispatched first), the rest the
// dgrad tiles.  Measured before this (rocprofv3 kernel trace of the captured step): the two chains on two streams
// overlapped little -- every cross-queue dependency of the graph cost 5-10 us.
#include "c1v_gemm_dev.h"
#include <stdlib.h>

struct PairParams {
    NtParams nt;
    TnParams tn;
    int tn_gi, tn_gij, tn_blocks, nt_gm;
    int tn_xcd, tn_gj, tn_groups;      // XCD-aware deal of the wgrad workgroups (pair_tn_tile)
};

// Which wgrad tile workgroup `id` of the grid runs.  Plain deal: id -> (tap * splits + split, by, bx), tile index fastest.
// XCD-aware deal (tn_xcd): workgroups go to the XCDs round-robin (XCD = id % 8), and the workgroups that read the same rows
// of the same operand columns -- the gj column tiles x taps of one (row split, bx) -- are all given to ONE XCD, so that
// its L2 fetches those rows once instead of every XCD fetching them (per-launch FETCH_SIZE of the paired launches at C4
// showed 1.5-2.3 x the operand bytes).  Groups are dealt to the XCDs round-robin; the grid is padded to whole rounds of
// eight groups, the padding workgroups return false.
__device__ __forceinline__ bool pair_tn_tile(const PairParams& pp, int id, int& bx, int& by, int& bz) {
    if (!pp.tn_xcd) {
        bz = id / pp.tn_gij;
        const int r = id - bz * pp.tn_gij;
        by = r / pp.tn_gi; bx = r - by * pp.tn_gi;
        return true;
    }
    const int grp = pp.tn_gj * pp.tn.taps;
    const int xcd = id & 7, slot = id >> 3;
    const int g = xcd + 8 * (slot / grp), w = slot % grp;
    if (g >= pp.tn_groups) return false;
    const int split = g / pp.tn_gi, tap = w / pp.tn_gj;
    bx = g - split * pp.tn_gi; by = w - tap * pp.tn_gj;
    bz = tap * pp.tn.splits + split;
    return true;
}

template <typename T, int BM, int BN, int NS>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_pair_kernel(const PairParams pp) {
    static_assert(sizeof(PairParams) + 192 > 0x1c0, "kernarg lines touched below");
    c1v_touch_kernarg<7>();
    const int id = blockIdx.x;
    if (id < pp.tn_blocks) {
        int bx, by, bz;
        if (!pair_tn_tile(pp, id, bx, by, bz)) return;
        tn_body<T>(pp.tn, bx, by, bz);
    } else {
        const int r = id - pp.tn_blocks;
        const int by = r / pp.nt_gm, bx = r - by * pp.nt_gm;
        nt_body<T, BM, BN, NS>(pp.nt, bx, by, r);
    }
}

// large-GEMM variant (see nt_variant): two workgroups per CU
template <typename T, int NS>
__global__ __launch_bounds__(C1V_GEMM_THREADS, 3) void gemm_pair_large_kernel(const PairParams pp) {
    c1v_touch_kernarg<7>();
    const int id = blockIdx.x;
    if (id < pp.tn_blocks) {
        int bx, by, bz;
        if (!pair_tn_tile(pp, id, bx, by, bz)) return;
        tn_body<T>(pp.tn, bx, by, bz);
    } else {
        // (r % 8 names the XCD as well as id % 8 does: the same shift for every workgroup)
        const int r = id - pp.tn_blocks;
        int bx, by;
        xcd_tile(r, pp.nt_gm, (pp.nt.e.N + 127) >> 7, bx, by);
        nt_body<T, 128, 128, NS, false>(pp.nt, bx, by, r);
    }
}

// the same with the window variant of the dgrad tiles (ntw_body, 128 x 128, two workgroups per CU; one row mode per
// instantiation)
template <typename T, int MODE>
__global__ __launch_bounds__(C1V_GEMM_THREADS, 3) void gemm_pair_ntw_kernel(const PairParams pp) {
    c1v_touch_kernarg<7>();
    const int id = blockIdx.x;
    if (id < pp.tn_blocks) {
        int bx, by, bz;
        if (!pair_tn_tile(pp, id, bx, by, bz)) return;
        tn_body<T>(pp.tn, bx, by, bz);
    } else {
        const int r = id - pp.tn_blocks;
        int bx, by;
        xcd_tile(r, pp.nt_gm, (pp.nt.e.N + NtwTwo::BN - 1) / NtwTwo::BN, bx, by);
        ntw_body<T, MODE, NtwTwo>(pp.nt, bx, by);
    }
}
template <typename T, int MODE> static void launch_pair_ntw_m(PairParams& pp, hipStream_t st) {
    constexpr int LDS = NtwTwo::LDS > TN_LDS ? NtwTwo::LDS : TN_LDS;
    static bool once = (set_lds(gemm_pair_ntw_kernel<T, MODE>, LDS), true); (void)once;
    pp.nt_gm = (int)c1v_cdiv(pp.nt.e.M, NtwTwo::BM);
    const int64_t nt_blocks = (int64_t)pp.nt_gm * c1v_cdiv(pp.nt.e.N, NtwTwo::BN);
    hipLaunchKernelGGL((gemm_pair_ntw_kernel<T, MODE>), dim3((unsigned)(pp.tn_blocks + nt_blocks)), dim3(C1V_GEMM_THREADS), LDS, st, pp);
}
template <typename T> static void launch_pair_ntw(PairParams& pp, hipStream_t st) {
    switch (pp.nt.e.row_mode) {
        case C1V_ROW_STORE:   launch_pair_ntw_m<T, C1V_ROW_STORE>(pp, st); break;
        case C1V_ROW_POOL2:   launch_pair_ntw_m<T, C1V_ROW_POOL2>(pp, st); break;
        case C1V_ROW_UP2:     launch_pair_ntw_m<T, C1V_ROW_UP2>(pp, st); break;
        case C1V_ROW_PAIRSUM: launch_pair_ntw_m<T, C1V_ROW_PAIRSUM>(pp, st); break;
        default:              launch_pair_ntw_m<T, C1V_ROW_UNPOOL>(pp, st); break;
    }
}

template <typename T, int NS> static void launch_pair_large_ns(PairParams& pp, hipStream_t st) {
    constexpr int NT_LDS = nt_lds<128, 128, NS>();
    constexpr int LDS = NT_LDS > TN_LDS ? NT_LDS : TN_LDS;
    static bool once = (set_lds(gemm_pair_large_kernel<T, NS>, LDS), true); (void)once;
    pp.nt_gm = (int)c1v_cdiv(pp.nt.e.M, 128);
    const int64_t nt_blocks = (int64_t)pp.nt_gm * c1v_cdiv(pp.nt.e.N, 128);
    hipLaunchKernelGGL((gemm_pair_large_kernel<T, NS>), dim3((unsigned)(pp.tn_blocks + nt_blocks)), dim3(C1V_GEMM_THREADS), LDS, st, pp);
}
template <typename T> static void launch_pair_large(PairParams& pp, hipStream_t st) {
    switch (nt_large_ns()) {
        case 4:  launch_pair_large_ns<T, 4>(pp, st); break;
        case 5:  launch_pair_large_ns<T, 5>(pp, st); break;
        default: launch_pair_large_ns<T, 2>(pp, st); break;
    }
}

template <typename T, int BM, int BN, int NS> static void launch_pair_v(PairParams& pp, hipStream_t st) {
    constexpr int NT_LDS = nt_lds<BM, BN, NS>();
    constexpr int LDS = NT_LDS > TN_LDS ? NT_LDS : TN_LDS;
    static bool once = (set_lds(gemm_pair_kernel<T, BM, BN, NS>, LDS), true); (void)once;
    pp.nt_gm = (int)c1v_cdiv(pp.nt.e.M, BM);
    const int64_t nt_blocks = (int64_t)pp.nt_gm * c1v_cdiv(pp.nt.e.N, BN);
    hipLaunchKernelGGL((gemm_pair_kernel<T, BM, BN, NS>), dim3((unsigned)(pp.tn_blocks + nt_blocks)), dim3(C1V_GEMM_THREADS), LDS, st, pp);
}

template <typename T> static int launch_pair_t(PairParams& pp, hipStream_t st) {
    if (pp.nt.wide == 1) {
        launch_pair_ntw<T>(pp, st);
        C1V_LAUNCH_CHECK();
        return 0;
    }
    switch (nt_variant(pp.nt)) {
        case NT_128L:    launch_pair_large<T>(pp, st); break;
        case NT_128:     launch_pair_v<T, 128, 128, 3>(pp, st); break;
        default:         launch_pair_v<T, 64, 64, 4>(pp, st); break;
    }
    C1V_LAUNCH_CHECK();
    return 0;
}

int c1v_launch_pair(int dtype, const NtParams& nt, const TnParams& tn, hipStream_t st) {
    if (c1v_check_nt(nt) || c1v_check_tn(tn)) return -1;
    C1V_REQUIRE(nt.ksplits <= 1, "pair launch: split-K dgrad is not supported");
    if (nt.wide >= 2 || tn.tg >= 2) {
        // the large-batch variants (one workgroup per CU each: 256 x 128 dgrad tiles, fused-tap wgrad) fill the chip by
        // themselves: two launches, wgrad first
        if (c1v_launch_tn(dtype, tn, st)) return -1;
        return c1v_launch_nt(dtype, nt, st);
    }
    PairParams pp;
    pp.nt = nt; pp.tn = tn;
    pp.tn_gi = (int)c1v_cdiv(tn.I, 128);
    pp.tn_gij = pp.tn_gi * (int)c1v_cdiv(tn.J, 128);
    int64_t tn_blocks = (int64_t)pp.tn_gij * tn.taps * tn.splits;
    pp.tn_gj = pp.tn_gij / pp.tn_gi;
    pp.tn_groups = tn.splits * pp.tn_gi;
    // the XCD-aware deal pays when there is something to share (several workgroups per group) and enough groups to keep the
    // eight XCDs evenly loaded (C1V_TN_XCD=0 / 1: never / always)
    const char* env = getenv("C1V_TN_XCD");
    pp.tn_xcd = env ? (env[0] == '1') : (pp.tn_gj * tn.taps >= 2 && pp.tn_groups >= 16);
    if (pp.tn_xcd) tn_blocks = 8 * c1v_cdiv(pp.tn_groups, 8) * pp.tn_gj * tn.taps;
    C1V_REQUIRE(tn_blocks < (1ll << 30), "pair launch: too many wgrad tiles");
    pp.tn_blocks = (int)tn_blocks;
    return dtype == C1V_F32 ? launch_pair_t<float>(pp, st) : launch_pair_t<bf16_raw>(pp, st);
}
