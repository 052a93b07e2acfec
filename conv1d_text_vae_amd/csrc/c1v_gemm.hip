// Kernels and host launchers of the two GEMM engines (device code: c1v_gemm_dev.h).
#include "c1v_gemm_dev.h"
#include <stdlib.h>

template <typename T, int BM, int BN, int NS>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_nt_kernel(const NtParams p) {
    static_assert(sizeof(NtParams) + 192 > 0x100, "kernarg lines touched below");
    c1v_touch_kernarg<4>();
    nt_body<T, BM, BN, NS>(p, blockIdx.x, blockIdx.y, blockIdx.x + gridDim.x * blockIdx.y, blockIdx.z);
}
// large-GEMM variant: two workgroups per CU (12 waves = 3 per SIMD)
template <typename T, int NS>
__global__ __launch_bounds__(C1V_GEMM_THREADS, 3) void gemm_nt_large_kernel(const NtParams p) {
    c1v_touch_kernarg<4>();
    int bx, by;
    xcd_tile(blockIdx.x, (p.e.M + 127) >> 7, (p.e.N + 127) >> 7, bx, by);
    nt_body<T, 128, 128, NS, false>(p, bx, by, blockIdx.x, 0);
}
// window variants (ntw_body): a linear grid dealt to (row tile, column tile) by xcd_tile()
template <typename T, int MODE>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_ntw_kernel(const NtParams p) {
    c1v_touch_kernarg<4>();
    int bx, by;
    xcd_tile(blockIdx.x, (p.e.M + NtwBig::BM - 1) / NtwBig::BM, (p.e.N + NtwBig::BN - 1) / NtwBig::BN, bx, by);
    ntw_body<T, MODE, NtwBig>(p, bx, by);
}
template <typename T, int MODE>
__global__ __launch_bounds__(C1V_GEMM_THREADS, 3) void gemm_ntw2_kernel(const NtParams p) {
    c1v_touch_kernarg<4>();
    int bx, by;
    xcd_tile(blockIdx.x, (p.e.M + NtwTwo::BM - 1) / NtwTwo::BM, (p.e.N + NtwTwo::BN - 1) / NtwTwo::BN, bx, by);
    ntw_body<T, MODE, NtwTwo>(p, bx, by);
}
// whole output rows + l2 normalisation (NtwRow / C1V_ROW_NORM): one column tile, row tiles in launch order
template <typename T, typename Cfg>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_ntw_row_kernel(const NtParams p) {
    c1v_touch_kernarg<4>();
    ntw_body<T, C1V_ROW_NORM, Cfg>(p, blockIdx.x, 0);
}
template <typename T>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_tn_kernel(const TnParams p) {
    tn_body<T>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several weight gradients in ONE launch (c1v_wgrad_batch): small batches, where every wgrad is a few dozen workgroups and
// its launch is as long as its longest workgroup.  A linear grid; problem k owns workgroups [first[k], first[k+1]).
struct TnMulti {
    int n;
    int first[C1V_TN_MULTI_MAX + 1], gi[C1V_TN_MULTI_MAX], gij[C1V_TN_MULTI_MAX];
    TnParams p[C1V_TN_MULTI_MAX];
};
template <typename T, int NS>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_tn_multi_kernel(const TnMulti m) {
    const int id = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int i = 1; i < C1V_TN_MULTI_MAX; ++i)
        if (i < m.n && id >= m.first[i]) k = i;
    k = __builtin_amdgcn_readfirstlane(k);
    const int local = id - m.first[k], gi = m.gi[k], gij = m.gij[k];
    const int bz = local / gij, r = local - bz * gij;
    const int by = r / gi, bx = r - by * gi;
    if constexpr (NS >= 4) {
        // (the deep-ring form has the LDS for the fused-tap body as well; bz = tap group * splits + split there)
        const int tg = __builtin_amdgcn_readfirstlane(m.p[k].tg);
        if (tg == 3) { tn_taps_body<T, 3>(m.p[k], bx, by, bz); return; }
        if (tg == 2) { tn_taps_body<T, 2>(m.p[k], bx, by, bz); return; }
    }
    tn_body<T, NS>(m.p[k], bx, by, bz);
}

// Fused-tap wgrad: a linear grid, dealt to (tile, tap group, row split) so that the workgroups which read the SAME rows --
// the 128 x 128 tiles and tap groups of one row split -- are dispatched back to back to ONE XCD (workgroups go to the XCDs
// round-robin; XCD k = id % 8 owns the row splits k, k + 8, ...) and share its L2 copy of both operands.  With the plain
// 3-d grid the tiles of a split landed on different XCDs and every one of them fetched its rows from HBM again: measured
// at C4 (rocprofv3 FETCH_SIZE per launch), 2-4 x the operand bytes, at 4.0-4.6 TB/s -- the launches were HBM-bound on re-reads.
template <typename T, int TG>
__global__ __launch_bounds__(C1V_GEMM_THREADS) void gemm_tn_taps_kernel(const TnParams p, const int gi, const int gj, const int groups) {
    const int per_split = gi * gj * groups;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int split = xcd + 8 * (slot / per_split);
    if (split >= p.splits) return;                       // (the grid is rounded up to whole rounds of 8 splits)
    int r = slot % per_split;
    const int tgi = r / (gi * gj);
    r -= tgi * gi * gj;
    const int by = r / gi, bx = r - by * gi;
    tn_taps_body<T, TG>(p, bx, by, tgi * p.splits + split);
}

// ------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int NS> static void launch_nt_v(const NtParams& p, hipStream_t st) {
    constexpr int LDS = nt_lds<BM, BN, NS>();
    static bool once = (set_lds(gemm_nt_kernel<T, BM, BN, NS>, LDS), true); (void)once;
    dim3 grid((unsigned)c1v_cdiv(p.e.M, BM), (unsigned)c1v_cdiv(p.e.N, BN), (unsigned)(p.ksplits > 1 ? p.ksplits : 1));
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, NS>), grid, dim3(C1V_GEMM_THREADS), LDS, st, p);
}

template <typename T, int NS> static void launch_nt_large_ns(const NtParams& p, hipStream_t st) {
    constexpr int LDS = nt_lds<128, 128, NS>();
    static bool once = (set_lds(gemm_nt_large_kernel<T, NS>, LDS), true); (void)once;
    // a linear grid, dealt to (row tile, column tile) by xcd_tile()
    dim3 grid((unsigned)(c1v_cdiv(p.e.M, 128) * c1v_cdiv(p.e.N, 128)), 1, 1);
    hipLaunchKernelGGL((gemm_nt_large_kernel<T, NS>), grid, dim3(C1V_GEMM_THREADS), LDS, st, p);
}
template <typename T> static void launch_nt_large(const NtParams& p, hipStream_t st) {
    switch (nt_large_ns()) {
        case 4:  launch_nt_large_ns<T, 4>(p, st); break;
        case 5:  launch_nt_large_ns<T, 5>(p, st); break;
        default: launch_nt_large_ns<T, 2>(p, st); break;
    }
}

template <typename T, int MODE> static void launch_ntw_m(const NtParams& p, hipStream_t st) {
    if (p.wide == 2) {
        static bool once = (set_lds(gemm_ntw_kernel<T, MODE>, NtwBig::LDS), true); (void)once;
        dim3 grid((unsigned)(c1v_cdiv(p.e.M, NtwBig::BM) * c1v_cdiv(p.e.N, NtwBig::BN)), 1, 1);
        hipLaunchKernelGGL((gemm_ntw_kernel<T, MODE>), grid, dim3(C1V_GEMM_THREADS), NtwBig::LDS, st, p);
    } else {
        static bool once = (set_lds(gemm_ntw2_kernel<T, MODE>, NtwTwo::LDS), true); (void)once;
        dim3 grid((unsigned)(c1v_cdiv(p.e.M, NtwTwo::BM) * c1v_cdiv(p.e.N, NtwTwo::BN)), 1, 1);
        hipLaunchKernelGGL((gemm_ntw2_kernel<T, MODE>), grid, dim3(C1V_GEMM_THREADS), NtwTwo::LDS, st, p);
    }
}
template <typename T> static void launch_ntw(const NtParams& p, hipStream_t st) {
    if (p.wide == 3) {
        // fewer 128-row tiles than CUs: 64-row tiles instead (twice the workgroups, each stores half as much; the launch
        // is one round of tiles either way and its length is the stores of one workgroup)
        if (c1v_cdiv(p.e.M, NtwRow::BM) < 256) {
            static bool once = (set_lds(gemm_ntw_row_kernel<T, NtwRowS>, NtwRowS::LDS), true); (void)once;
            hipLaunchKernelGGL((gemm_ntw_row_kernel<T, NtwRowS>), dim3((unsigned)c1v_cdiv(p.e.M, NtwRowS::BM)), dim3(C1V_GEMM_THREADS), NtwRowS::LDS, st, p);
        } else {
            static bool once = (set_lds(gemm_ntw_row_kernel<T, NtwRow>, NtwRow::LDS), true); (void)once;
            hipLaunchKernelGGL((gemm_ntw_row_kernel<T, NtwRow>), dim3((unsigned)c1v_cdiv(p.e.M, NtwRow::BM)), dim3(C1V_GEMM_THREADS), NtwRow::LDS, st, p);
        }
        return;
    }
    switch (p.e.row_mode) {
        case C1V_ROW_STORE:   launch_ntw_m<T, C1V_ROW_STORE>(p, st); break;
        case C1V_ROW_POOL2:   launch_ntw_m<T, C1V_ROW_POOL2>(p, st); break;
        case C1V_ROW_UP2:     launch_ntw_m<T, C1V_ROW_UP2>(p, st); break;
        case C1V_ROW_PAIRSUM: launch_ntw_m<T, C1V_ROW_PAIRSUM>(p, st); break;
        default:              launch_ntw_m<T, C1V_ROW_UNPOOL>(p, st); break;
    }
}

template <typename T> static int launch_nt_t(const NtParams& p, hipStream_t st) {
    if (p.wide) {
        launch_ntw<T>(p, st);
        C1V_LAUNCH_CHECK();
        return 0;
    }
    switch (nt_variant(p)) {
        case NT_128L:    launch_nt_large<T>(p, st); break;
        case NT_128:     launch_nt_v<T, 128, 128, 3>(p, st); break;
        default:         launch_nt_v<T, 64, 64, 4>(p, st); break;
    }
    C1V_LAUNCH_CHECK();
    return 0;
}

int c1v_check_nt(const NtParams& p) {
    C1V_REQUIRE(p.e.M > 0 && p.e.N > 0 && p.Kb > 0, "empty GEMM (M=%d N=%d Kb=%lld)", p.e.M, p.e.N, (long long)p.Kb);
    C1V_REQUIRE(p.Kb % 16 == 0 && p.lda_b % 16 == 0 && p.ldw_b % 16 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0,
                "GEMM operands must be 16-byte aligned (Kb=%lld lda=%lld ldw=%lld)", (long long)p.Kb, (long long)p.lda_b, (long long)p.ldw_b);
    C1V_REQUIRE(p.ldw_b >= c1v_cdiv(p.Kb, 128) * 128, "weight operand rows must be zero-padded to whole 128-byte K tiles (ldw=%lld bytes, K=%lld bytes)",
                (long long)p.ldw_b, (long long)p.Kb);
    const bool pairs = p.e.row_mode == C1V_ROW_POOL2 || p.e.row_mode == C1V_ROW_PAIRSUM;
    C1V_REQUIRE(!pairs || (p.e.Le % 2 == 0), "pair epilogue needs an even row enumeration (Le=%d)", p.e.Le);
    if (p.wide)
        C1V_REQUIRE(p.k_taps >= 2 && p.k_taps <= 9 && p.k_strips >= 1 && p.Kb == (int64_t)p.k_taps * p.k_strips * 128 && p.ksplits <= 1 &&
                    p.e.row_mode != C1V_ROW_LATENT_BWD && (p.wide == 3 ? p.e.N <= NtwRow::BN : p.e.fastw == 4),
                    "wide NT variant: needs a convolution over whole 128-byte channel strips (taps %d, strips %d, K %lld bytes)",
                    p.k_taps, p.k_strips, (long long)p.Kb);
    C1V_REQUIRE((p.e.row_mode == C1V_ROW_NORM) == (p.wide == 3), "ROW_NORM comes with the whole-row variant only (convolutions over whole 128-byte channel strips)");
    if (p.ksplits > 1) {
        C1V_REQUIRE(p.e.row_mode == C1V_ROW_STORE && !p.e.relu && !p.e.exp2 && p.e.act == nullptr && p.e.keep == nullptr,
                    "split-K needs a plain STORE epilogue (partial sums are not final values)");
        C1V_REQUIRE(p.kt_per_split > 0 && (int64_t)p.kt_per_split * p.ksplits >= c1v_cdiv(p.Kb, 128), "split-K ranges do not cover K");
    }
    return 0;
}

int c1v_check_tn(const TnParams& p) {
    C1V_REQUIRE(p.I > 0 && p.J > 0 && p.taps > 0 && p.splits > 0, "empty wgrad");
    C1V_REQUIRE(p.lda_b % 16 == 0 && p.ldd_b % 16 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.D % 16) == 0,
                "wgrad operands must be 16-byte aligned");
    return 0;
}

int c1v_launch_nt(int dtype, const NtParams& p, hipStream_t st) {
    if (c1v_check_nt(p)) return -1;
    return dtype == C1V_F32 ? launch_nt_t<float>(p, st) : launch_nt_t<bf16_raw>(p, st);
}

template <typename T, int TG> static void launch_tn_taps(const TnParams& p, hipStream_t st) {
    static bool once = (set_lds(gemm_tn_taps_kernel<T, TG>, TNF_LDS), true); (void)once;
    const int gi = (int)c1v_cdiv(p.I, 128), gj = (int)c1v_cdiv(p.J, 128), groups = (int)c1v_cdiv(p.taps, TG);
    const int64_t blocks = c1v_cdiv(p.splits, 8) * 8 * gi * gj * groups;
    hipLaunchKernelGGL((gemm_tn_taps_kernel<T, TG>), dim3((unsigned)blocks), dim3(C1V_GEMM_THREADS), TNF_LDS, st, p, gi, gj, groups);
}

template <typename T> static int launch_tn_t(const TnParams& p, hipStream_t st) {
    if (p.tg >= 2) {
        if (p.tg == 2) launch_tn_taps<T, 2>(p, st); else launch_tn_taps<T, 3>(p, st);
        C1V_LAUNCH_CHECK();
        return 0;
    }
    static bool once = (set_lds(gemm_tn_kernel<T>, TN_LDS), true); (void)once;
    dim3 grid((unsigned)c1v_cdiv(p.I, 128), (unsigned)c1v_cdiv(p.J, 128), (unsigned)(p.taps * p.splits));
    hipLaunchKernelGGL((gemm_tn_kernel<T>), grid, dim3(C1V_GEMM_THREADS), TN_LDS, st, p);
    C1V_LAUNCH_CHECK();
    return 0;
}

int c1v_launch_tn(int dtype, const TnParams& p, hipStream_t st) {
    if (c1v_check_tn(p)) return -1;
    return dtype == C1V_F32 ? launch_tn_t<float>(p, st) : launch_tn_t<bf16_raw>(p, st);
}

int c1v_launch_tn_multi(int dtype, const TnParams* ps, int n, hipStream_t st) {
    C1V_REQUIRE(ps && n >= 1 && n <= C1V_TN_MULTI_MAX, "wgrad batch: 1..%d problems", C1V_TN_MULTI_MAX);
    TnMulti m{};
    m.n = n;
    int64_t blocks = 0;
    bool fused = false;
    for (int k = 0; k < n; ++k) {
        if (c1v_check_tn(ps[k])) return -1;
        m.p[k] = ps[k];
        m.gi[k] = (int)c1v_cdiv(ps[k].I, 128);
        m.gij[k] = m.gi[k] * (int)c1v_cdiv(ps[k].J, 128);
        m.first[k] = (int)blocks;
        const int groups = ps[k].tg >= 2 ? ps[k].taps / ps[k].tg : ps[k].taps;       // (tg divides taps: wgrad_params)
        fused = fused || ps[k].tg >= 2;
        blocks += (int64_t)m.gij[k] * groups * ps[k].splits;
        C1V_REQUIRE(blocks < (1ll << 30), "wgrad batch: too many tiles");
    }
    for (int k = n; k <= C1V_TN_MULTI_MAX; ++k) m.first[k] = (int)blocks;
    // at most one workgroup per CU: the deep ring (144 KiB, one workgroup per CU anyway); more: two per CU with the shallow one
    const char* env = getenv("C1V_TN_MULTI_NS");
    const bool deep = fused || (env ? env[0] == '4' : blocks <= 256);        // (fused-tap problems need the deep form's LDS)
    constexpr int LDS_DEEP = 4 * 2 * TN_TILE > TNF_LDS ? 4 * 2 * TN_TILE : TNF_LDS;
#define C1V_TN_MULTI(T_, NS_, LDS_) do { \
        static bool once = (set_lds(gemm_tn_multi_kernel<T_, NS_>, LDS_), true); (void)once; \
        hipLaunchKernelGGL((gemm_tn_multi_kernel<T_, NS_>), dim3((unsigned)blocks), dim3(C1V_GEMM_THREADS), LDS_, st, m); } while (0)
    if (dtype == C1V_F32) { if (deep) C1V_TN_MULTI(float, 4, LDS_DEEP); else C1V_TN_MULTI(float, 2, TN_LDS); }
    else { if (deep) C1V_TN_MULTI(bf16_raw, 4, LDS_DEEP); else C1V_TN_MULTI(bf16_raw, 2, TN_LDS); }
#undef C1V_TN_MULTI
    C1V_LAUNCH_CHECK();
    return 0;
}

int c1v_fill_epi(const c1v_epilogue& e, EpiDev& d, int M, int Le, int nrows, int N) {
    d.bias = e.bias; d.relu = e.relu; d.row_mode = e.row_mode; d.reshape_w = e.reshape_w; d.exp2 = 0;
    d.dst = e.dst.data; d.dst_dtype = e.dst.dtype; d.dst_bstride = e.dst.bstride;
    d.dst_off = e.dst.off; d.dst_len = e.dst.len; d.dst_cp = e.dst.cp; d.dst_c = e.dst.c;
    d.act = e.act.data; d.act_dtype = e.act.dtype; d.act_bstride = e.act.bstride; d.act_off = e.act.off;
    d.act_cp = e.act.cp; d.act_rowmul = e.act_rowmul > 0 ? e.act_rowmul : 1; d.act_scale = e.act_scale;
    d.sel = e.sel; d.sel_bstride = e.sel_bstride; d.sel_cp = e.sel_cp;
    d.keep = e.keep; d.keep_scale = e.keep_scale; d.bias_row_w = e.bias_row_w;
    d.lat_mlv = e.lat_mlv; d.lat_eps = e.lat_eps; d.lat_keep = e.lat_keep; d.lat_D = e.lat_D;
    d.lat_keep_prob = e.lat_keep_prob; d.lat_std_scale = e.lat_std_scale; d.lat_kc = e.lat_kc;
    d.M = M; d.Le = Le; d.nrows = nrows; d.N = N;
    d.nrm_yhat = e.nrm_yhat; d.nrm_kp = e.nrm_kp; d.nrm_rnorm = e.nrm_rnorm; d.nrm_qscale = e.nrm_qscale; d.nrm_pos = e.nrm_pos;
    d.nrm_live_only = e.nrm_live_only; d.nrm_T = 0;
    C1V_REQUIRE(e.dst.data != nullptr, "epilogue has no destination");
    C1V_REQUIRE(e.row_mode >= C1V_ROW_STORE && e.row_mode <= C1V_ROW_NORM, "bad row_mode %d", e.row_mode);
    if (e.row_mode == C1V_ROW_NORM) {
        C1V_REQUIRE(e.nrm_yhat && e.nrm_rnorm && e.nrm_kp >= N && e.nrm_kp % 8 == 0 && e.nrm_kp <= 320, "ROW_NORM: needs yhat, rnorm and n_out <= nrm_kp <= 320");
        C1V_REQUIRE(!e.relu && e.act.data == nullptr && e.keep == nullptr && e.sel == nullptr && e.reshape_w <= 0 && e.dst.dtype == C1V_F32,
                    "ROW_NORM: a plain fp32 STORE epilogue (bias only)");
        d.nrm_dy = nullptr;
        if (e.nrm_dy.data) {
            const int es = c1v_esize(e.nrm_dy.dtype);
            C1V_REQUIRE(e.nrm_pos && e.nrm_dy.len >= nrows && e.nrm_dy.batch * (int64_t)1 > 0 && (e.nrm_dy.cp * es) % 16 == 0 && e.nrm_dy.cp * es <= 1024 &&
                        (e.nrm_dy.bstride * es) % 16 == 0 && ((uintptr_t)e.nrm_dy.data % 16) == 0,
                        "ROW_NORM: nrm_dy needs nrm_pos and rows of whole 16-byte words, at most 1 KiB each");
            d.nrm_dy = (char*)e.nrm_dy.data; d.nrm_dy_bstride_b = e.nrm_dy.bstride * es; d.nrm_dy_off = e.nrm_dy.off; d.nrm_dy_row_b = e.nrm_dy.cp * es;
        }
        C1V_REQUIRE(e.dst.cp <= 320 && e.dst.cp % 4 == 0 && e.dst.bstride % 4 == 0 && ((uintptr_t)e.dst.data % 16) == 0 && ((uintptr_t)e.nrm_yhat % 16) == 0,
                    "ROW_NORM: rows of dst and yhat must be 16-byte aligned (cp %d)", e.dst.cp);
        d.fastw = 0;
        return 0;
    }
    C1V_REQUIRE(e.row_mode != C1V_ROW_UNPOOL || e.sel != nullptr, "UNPOOL needs sel");
    C1V_REQUIRE(e.reshape_w <= 0 || (e.row_mode == C1V_ROW_STORE || e.row_mode == C1V_ROW_UNPOOL), "reshape_w only with STORE/UNPOOL");
    C1V_REQUIRE(e.bias_row_w == 0 || (e.reshape_w > 0 && e.bias_row_w > 0 && e.bias_row_w <= e.reshape_w), "bias_row_w %d needs reshape_w >= it", e.bias_row_w);
    const int c_max = e.reshape_w > 0 ? e.reshape_w : N;
    C1V_REQUIRE(c_max <= e.dst.cp, "destination row pitch %d is smaller than %d channels", e.dst.cp, c_max);
    // widest vector width at which every 4-channel group of the epilogue stays aligned (see epi_load / epi_finish)
    d.fastw = 0;
    const bool masks = e.sel != nullptr || e.keep != nullptr;
    for (int vw = 4; vw >= 2 && d.fastw == 0; vw -= 2) {
        auto al = [&](const void* p, int esize) { return ((uintptr_t)p % (uintptr_t)(vw * esize)) == 0; };
        // (a plain fp32/bf16 STORE whose rows are padded to whole 4-channel groups may end in a partial group: the group's
        // surplus values land in the row's padding)
        const bool pad_tail = vw == 4 && e.row_mode == C1V_ROW_STORE && e.reshape_w <= 0 && e.act.data == nullptr && !masks &&
                              e.dst.cp >= (N + 3) / 4 * 4;
        bool ok = (N % vw == 0 || pad_tail) && (e.reshape_w <= 0 || e.reshape_w % 4 == 0);
        ok = ok && al(e.dst.data, c1v_esize(e.dst.dtype)) && e.dst.bstride % vw == 0 && e.dst.cp % vw == 0;
        ok = ok && (e.bias == nullptr || al(e.bias, 4));
        if (e.act.data) ok = ok && al(e.act.data, c1v_esize(e.act.dtype)) && e.act.bstride % vw == 0 && e.act.cp % vw == 0;
        if (masks) {
            const int64_t n_samples = c1v_cdiv(M, Le > 0 ? Le : 1);
            ok = ok && e.sel_bstride % vw == 0 && e.sel_cp % vw == 0 && n_samples * e.sel_bstride + e.sel_cp < (int64_t)1 << 31;
            ok = ok && (e.sel == nullptr || al(e.sel, 1)) && (e.keep == nullptr || al(e.keep, 1));
        }
        if (e.row_mode == C1V_ROW_LATENT_BWD)
            ok = ok && vw == 4 && e.lat_D % 4 == 0 && al(e.lat_mlv, 4) && al(e.lat_eps, 4) && (e.lat_keep == nullptr || al(e.lat_keep, 1));
        if (ok) d.fastw = vw;
    }
    if (e.row_mode == C1V_ROW_LATENT_BWD) {
        C1V_REQUIRE(e.lat_mlv && e.lat_eps && e.lat_D == N && e.lat_keep_prob > 0.f, "LATENT_BWD: needs lat_mlv, lat_eps and lat_D == N");
        C1V_REQUIRE(e.bias == nullptr && !e.relu && e.act.data == nullptr && e.keep == nullptr && e.reshape_w <= 0, "LATENT_BWD: no bias / relu / act / keep / reshape");
        C1V_REQUIRE(e.dst.cp >= 2 * N, "LATENT_BWD: destination rows must hold 2*lat_D values");
        C1V_REQUIRE(d.fastw == 4, "LATENT_BWD needs the 4-wide vector epilogue (lat_D %% 4 == 0, 16-byte aligned buffers)");
    }
    if (const char* env = getenv("C1V_EPI_ELEMENTWISE")) if (env[0] == '1') d.fastw = 0;     // diagnostic: force the element-wise path
    if (getenv("C1V_EPI_TRACE")) fprintf(stderr, "c1v epilogue: mode %d M %d N %d reshape_w %d dst_cp %d -> vector width %d\n", e.row_mode, M, N, e.reshape_w, e.dst.cp, d.fastw);
    return 0;
}
