// C-ABI entry points that map the layer-level ops (conv / deconv / dense forward, dgrad, wgrad) onto the GEMM
// engine of c1v_gemm.hip, plus error / version / device-info plumbing.  Every shape is checked here, on the host,
// before anything is launched.
#include <stdarg.h>
#include <stdlib.h>
#include "c1v_common.h"

static thread_local std::string g_err;

void c1v_set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char* c1v_last_error(void) { return g_err.c_str(); }
extern "C" int c1v_version(void) { return 100; }

extern "C" int c1v_device_info(int32_t device, int32_t* info4) {
    C1V_REQUIRE(info4, "device_info: null argument");
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    C1V_REQUIRE(e == hipSuccess, "device_info: %s", hipGetErrorString(e));
    info4[0] = prop.multiProcessorCount;
    info4[1] = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    info4[2] = prop.warpSize;
    int arch = 0;
    sscanf(prop.gcnArchName, "gfx%d", &arch);
    info4[3] = arch;
    return 0;
}

static int check_operand_seq(const c1v_seq& s, const char* what) {
    C1V_REQUIRE(s.data != nullptr, "%s: null buffer", what);
    C1V_REQUIRE(s.dtype == C1V_F32 || s.dtype == C1V_BF16, "%s: bad dtype %d", what, s.dtype);
    C1V_REQUIRE(s.batch > 0 && s.len > 0 && s.off >= 0 && s.c > 0 && s.c <= s.cp, "%s: bad descriptor", what);
    C1V_REQUIRE(s.cp % 32 == 0, "%s: operand row pitch (%d) must be a multiple of 32 elements", what, s.cp);
    C1V_REQUIRE(s.bstride % s.cp == 0 && s.bstride >= (int64_t)(s.off + s.len) * s.cp, "%s: sample stride must be whole rows covering off+len", what);
    return 0;
}

// ---- conv / deconv forward and dgrad ------------------------------------------------------------------
static int conv_params(const c1v_conv1d_args* a, const char* what, NtParams& p) {
    C1V_REQUIRE(a && a->w_op, "%s: null argument", what);
    if (check_operand_seq(a->x, what)) return -1;
    C1V_REQUIRE(a->k > 0 && a->n_rows > 0 && a->n_out > 0 && a->a_off >= 0, "%s: bad shape k=%d n_rows=%d n_out=%d", what, a->k, a->n_rows, a->n_out);
    const int pitch = (int)(a->x.bstride / a->x.cp);
    // every window read must stay inside the allocation: the last sample may run into the slack rows
    C1V_REQUIRE(a->a_off + a->n_rows + a->k - 1 <= pitch + C1V_SLACK_ROWS, "%s: window (a_off=%d n_rows=%d k=%d) overruns the sample pitch %d", what, a->a_off, a->n_rows, a->k, pitch);
    const bool pairs = a->epi.row_mode == C1V_ROW_POOL2 || a->epi.row_mode == C1V_ROW_PAIRSUM;
    int Le = pairs ? ((a->n_rows + 1) & ~1) : a->n_rows;
    C1V_REQUIRE(a->a_off + Le + a->k - 1 <= pitch + C1V_SLACK_ROWS, "%s: padded row enumeration overruns the sample pitch", what);
    const int es = c1v_esize(a->x.dtype);
    p = NtParams{};
    p.A = (const char*)a->x.data; p.a_bstride_b = a->x.bstride * es; p.lda_b = (int64_t)a->x.cp * es; p.a_off = a->a_off;
    p.W = (const char*)a->w_op; p.ldw_b = a->ldw * es; p.Kb = (int64_t)a->k * a->x.cp * es;
    C1V_REQUIRE(a->ldw >= (int64_t)a->k * a->x.cp, "%s: ldw (%lld) is smaller than k*cp", what, (long long)a->ldw);
    if (a->k > 1 && ((int64_t)a->x.cp * es) % 128 == 0) { p.k_taps = a->k; p.k_strips = (int)((int64_t)a->x.cp * es / 128); }
    int64_t M = (int64_t)a->x.batch * Le;
    // Large convolutions: the wide variant (ntw_body) enumerates the FLAT buffer rows (Le = sample pitch; the rows between
    // the samples are computed and dropped by the epilogue) so that a tile's activation rows are one contiguous window,
    // fetched once per channel strip for all taps.  Taken when the launch fills the chip with two 128 x 128 workgroups per CU
    // and at most `1 - fill` of the enumerated rows are such gaps (C1V_NTW=0 disables, =1 forces wherever it applies).
    // C1V_NTW=2 / big selects the 256 x 128 one-workgroup-per-CU shape instead: measured at C4 (profiles/r03_c4_ntw_stamps.txt)
    // its K loop runs at 54 % MFMA utilisation, but the epilogues of all CUs then fall together -- an HBM write burst nothing
    // overlaps (19-38 k cycles per tile) -- and 4.06 rounds of tiles take 5.
    if (p.k_taps >= 2 && a->k <= 9 && pitch % 2 == 0 && a->x.bstride == (int64_t)pitch * a->x.cp &&
        a->epi.row_mode != C1V_ROW_LATENT_BWD) {
        const char* env = getenv("C1V_NTW");               // 0: off; 1 / 2: force the 128-row / 256-row shape; big: 256-row by size
        const char* env_t = getenv("C1V_NTW_MIN_TILES");
        const char* env_f = getenv("C1V_NTW_MIN_FILL");
        const int64_t Mf = (int64_t)a->x.batch * pitch;
        const int64_t tiles = c1v_cdiv(Mf, 128) * c1v_cdiv(a->n_out, 128);
        const bool off = env && env[0] == '0', force = env && (env[0] == '1' || env[0] == '2');
        const double fill = (double)a->n_rows / (double)pitch;
        const bool by_size = tiles >= (env_t ? atoll(env_t) : 512) && fill >= (env_f ? atof(env_f) : 0.85);
        if (!off && Mf < (1ll << 31) && (force || by_size)) {
            p.wide = (env && (env[0] == '2' || env[0] == 'b')) ? 2 : 1;
            Le = pitch;
            M = Mf;
        }
    }
    if (a->epi.row_mode == C1V_ROW_NORM) {                // whole rows in one workgroup: the flat-row enumeration at any size
        C1V_REQUIRE(p.k_taps >= 2 && a->k <= 9 && a->x.bstride == (int64_t)pitch * a->x.cp,
                    "%s: ROW_NORM needs 2..9 taps over whole 128-byte channel strips and samples one after the other", what);
        p.wide = 3;
        Le = pitch;
        M = (int64_t)a->x.batch * pitch;
    }
    C1V_REQUIRE(M < (1ll << 31), "%s: too many rows", what);
    if (c1v_fill_epi(a->epi, p.e, (int)M, Le, a->n_rows, a->n_out)) return -1;
    if (p.wide == 3) { p.e.nrm_T = a->x.dtype; return 0; }
    if (p.wide && p.e.fastw != 4) {                      // the wide variant only comes with the 4-wide vector epilogue
        p.wide = 0;
        Le = pairs ? ((a->n_rows + 1) & ~1) : a->n_rows;
        return c1v_fill_epi(a->epi, p.e, (int)((int64_t)a->x.batch * Le), Le, a->n_rows, a->n_out);
    }
    return 0;
}
static int conv_common(const c1v_conv1d_args* a, void* stream, const char* what) {
    NtParams p;
    if (conv_params(a, what, p)) return -1;
    return c1v_launch_nt(a->x.dtype, p, (hipStream_t)stream);
}
extern "C" int c1v_conv1d_fwd(const c1v_conv1d_args* a, void* s) { return conv_common(a, s, "conv1d_fwd"); }
extern "C" int c1v_conv1d_dgrad(const c1v_conv1d_args* a, void* s) { return conv_common(a, s, "conv1d_dgrad"); }
extern "C" int c1v_deconv1d_fwd(const c1v_conv1d_args* a, void* s) {
    C1V_REQUIRE(a, "deconv1d_fwd: null argument");
    C1V_REQUIRE(a->x.off == a->k - 1 && a->a_off == 0, "deconv1d_fwd: the input buffer must carry k-1 zero rows in front (off=%d, k=%d)", a->x.off, a->k);
    return conv_common(a, s, "deconv1d_fwd");
}
extern "C" int c1v_deconv1d_dgrad(const c1v_conv1d_args* a, void* s) { return conv_common(a, s, "deconv1d_dgrad"); }

// ---- dense forward / dgrad -----------------------------------------------------------------------------
static int dense_params(const c1v_dense_args* a, const char* what, NtParams& p) {
    C1V_REQUIRE(a && a->a && a->w_op, "%s: null argument", what);
    C1V_REQUIRE(a->dtype == C1V_F32 || a->dtype == C1V_BF16, "%s: bad dtype", what);
    C1V_REQUIRE(a->rows > 0 && a->K > 0 && a->n_out > 0 && a->lda >= a->K && a->ldw >= a->K, "%s: bad shape rows=%d K=%d n_out=%d", what, a->rows, a->K, a->n_out);
    const int es = c1v_esize(a->dtype);
    C1V_REQUIRE((a->K * es) % 16 == 0, "%s: K (%d) must be a whole number of 16-byte chunks", what, a->K);
    p = NtParams{};
    p.A = (const char*)a->a; p.a_bstride_b = a->lda * es; p.lda_b = a->lda * es; p.a_off = 0;
    p.W = (const char*)a->w_op; p.ldw_b = a->ldw * es; p.Kb = (int64_t)a->K * es;
    C1V_REQUIRE(a->epi.row_mode == C1V_ROW_STORE || a->epi.row_mode == C1V_ROW_UNPOOL || a->epi.row_mode == C1V_ROW_LATENT_BWD,
                "%s: dense epilogue supports STORE / UNPOOL / LATENT_BWD only", what);
    if (a->k_splits > 1) {
        const int64_t kt = c1v_cdiv(p.Kb, 128);
        C1V_REQUIRE(a->k_splits <= kt && a->split_stride > 0, "%s: bad split-K (k_splits=%d, K tiles=%lld)", what, a->k_splits, (long long)kt);
        p.ksplits = a->k_splits; p.kt_per_split = (int)c1v_cdiv(kt, a->k_splits); p.dst_split_stride = a->split_stride;
    }
    return c1v_fill_epi(a->epi, p.e, a->rows, 1, 1, a->n_out);
}
static int dense_common(const c1v_dense_args* a, void* stream, const char* what) {
    NtParams p;
    if (dense_params(a, what, p)) return -1;
    return c1v_launch_nt(a->dtype, p, (hipStream_t)stream);
}
extern "C" int c1v_dense_fwd(const c1v_dense_args* a, void* s) { return dense_common(a, s, "dense_fwd"); }
extern "C" int c1v_dense_dgrad(const c1v_dense_args* a, void* s) { return dense_common(a, s, "dense_dgrad"); }

// ---- wgrad -------------------------------------------------------------------------------------------------
static int wgrad_params(const c1v_wgrad_args* a, TnParams& p) {
    C1V_REQUIRE(a && a->a && a->d && a->out, "wgrad: null argument");
    C1V_REQUIRE(a->dtype == C1V_F32 || a->dtype == C1V_BF16, "wgrad: bad dtype");
    C1V_REQUIRE(a->I > 0 && a->J > 0 && a->I <= a->lda && a->J <= a->ldd, "wgrad: bad shape I=%d J=%d", a->I, a->J);
    C1V_REQUIRE(a->r_end > a->r_begin && a->taps > 0 && a->splits > 0, "wgrad: empty row range");
    C1V_REQUIRE(a->r_begin + a->a_shift0 >= 0, "wgrad: first A row would be negative");
    const int es = c1v_esize(a->dtype);
    const int per16 = 16 / es;
    p = TnParams{};
    p.A = (const char*)a->a; p.lda_b = a->lda * es; p.I = a->I; p.Ia = (int)(c1v_cdiv(a->I, per16) * per16);
    p.D = (const char*)a->d; p.ldd_b = a->ldd * es; p.J = a->J; p.Ja = (int)(c1v_cdiv(a->J, per16) * per16);
    C1V_REQUIRE(p.Ia <= a->lda && p.Ja <= a->ldd, "wgrad: rows must be padded to 16-byte chunks");
    p.r_begin = a->r_begin; p.r_end = a->r_end;
    const int br = a->dtype == C1V_F32 ? 32 : 64;
    p.rows_per_split = c1v_cdiv(c1v_cdiv(a->r_end - a->r_begin, a->splits), br) * br;
    p.a_shift0 = a->a_shift0; p.taps = a->taps; p.flip = a->flip; p.tap_stride = a->tap_stride; p.si = a->si; p.sj = a->sj;
    p.out = a->out; p.slab_stride = a->slab_stride; p.splits = a->splits;
    p.bias_out = a->bias_out; p.bias_slab_stride = a->bias_slab_stride;
    C1V_REQUIRE(a->tap_group >= 0 && a->tap_group <= 3, "wgrad: tap_group must be 0..3");
    p.tg = (a->tap_group >= 2 && a->taps >= 2) ? a->tap_group : 0;
    C1V_REQUIRE(p.tg == 0 || a->taps % p.tg == 0, "wgrad: tap_group (%d) must divide taps (%d)", a->tap_group, a->taps);
    return 0;
}
extern "C" int c1v_wgrad(const c1v_wgrad_args* a, void* stream) {
    TnParams p;
    if (wgrad_params(a, p)) return -1;
    return c1v_launch_tn(a->dtype, p, (hipStream_t)stream);
}

// ---- one backward stage in one launch: dgrad (conv / deconv / dense) + wgrad of the same dO buffer ------------
extern "C" int c1v_bwd_pair(const c1v_bwd_pair_args* a, void* stream) {
    C1V_REQUIRE(a && a->wgrad && ((a->conv_dgrad != nullptr) != (a->dense_dgrad != nullptr)), "bwd_pair: needs a wgrad and exactly one dgrad");
    NtParams nt;
    TnParams tn;
    int dtype;
    if (a->conv_dgrad) { if (conv_params(a->conv_dgrad, "bwd_pair dgrad", nt)) return -1; dtype = a->conv_dgrad->x.dtype; }
    else { if (dense_params(a->dense_dgrad, "bwd_pair dgrad", nt)) return -1; dtype = a->dense_dgrad->dtype; }
    if (wgrad_params(a->wgrad, tn)) return -1;
    C1V_REQUIRE(dtype == a->wgrad->dtype, "bwd_pair: dgrad and wgrad operand types differ");
    return c1v_launch_pair(dtype, nt, tn, (hipStream_t)stream);
}
extern "C" int c1v_conv1d_wgrad(const c1v_wgrad_args* a, void* s) { return c1v_wgrad(a, s); }
extern "C" int c1v_deconv1d_wgrad(const c1v_wgrad_args* a, void* s) { return c1v_wgrad(a, s); }
extern "C" int c1v_dense_wgrad(const c1v_wgrad_args* a, void* s) { return c1v_wgrad(a, s); }

// ---- several weight gradients in one launch ---------------------------------------------------------------
extern "C" int c1v_wgrad_batch(const c1v_wgrad_batch_args* a, void* stream) {
    C1V_REQUIRE(a && a->n >= 1 && a->n <= C1V_WGRAD_BATCH_MAX, "wgrad_batch: 1..%d problems", C1V_WGRAD_BATCH_MAX);
    static_assert(C1V_WGRAD_BATCH_MAX == C1V_TN_MULTI_MAX, "header and kernel agree on the batch size");
    TnParams ps[C1V_TN_MULTI_MAX];
    for (int k = 0; k < a->n; ++k) {
        C1V_REQUIRE(a->items[k], "wgrad_batch: item %d is null", k);
        if (wgrad_params(a->items[k], ps[k])) return -1;
        C1V_REQUIRE(a->items[k]->dtype == a->items[0]->dtype, "wgrad_batch: the problems' operand types differ");
    }
    return c1v_launch_tn_multi(a->items[0]->dtype, ps, a->n, (hipStream_t)stream);
}

// ---- recorded step --------------------------------------------------------------------------------------
extern "C" int c1v_run_list(int32_t n, const c1v_op_fn* fns, const void* const* args, void* stream) {
    C1V_REQUIRE(n >= 0 && (n == 0 || (fns && args)), "run_list: null argument");
    for (int32_t i = 0; i < n; ++i) {
        C1V_REQUIRE(fns[i] != nullptr, "run_list: entry %d has no function", i);
        const int rc = fns[i](args[i], stream);
        if (rc != 0) return rc;
    }
    return 0;
}

// The same with a second stream for work that is off the critical path (flags per entry: C1V_LIST_SIDE = issue on `side`;
// C1V_LIST_FORK = that side entry first waits for everything issued on `stream` so far; C1V_LIST_JOIN = that main entry first
// waits for everything issued on `side`).  If side work is still unjoined after the last entry, `stream` waits for it, so a
// list is always complete on `stream`.  Events: two per host thread and device, created on first use, never timed.
extern "C" int c1v_run_list_fork(int32_t n, const c1v_op_fn* fns, const void* const* args, const int32_t* flags, void* stream, void* side) {
    C1V_REQUIRE(n >= 0 && (n == 0 || (fns && args && flags)) && side && side != stream, "run_list_fork: bad argument");
    static thread_local hipEvent_t ev[16][2];
    static thread_local bool have[16];
    int dev = 0;
    C1V_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, "run_list_fork: no current device");
    if (!have[dev]) {
        for (int i = 0; i < 2; ++i) {
            const hipError_t e = hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming);
            C1V_REQUIRE(e == hipSuccess, "run_list_fork: hipEventCreate: %s", hipGetErrorString(e));
        }
        have[dev] = true;
    }
    hipStream_t sm = (hipStream_t)stream, ss = (hipStream_t)side;
    bool open_side = false;
    auto join = [&]() -> int {
        hipError_t e = hipEventRecord(ev[dev][1], ss);
        if (e == hipSuccess) e = hipStreamWaitEvent(sm, ev[dev][1], 0);
        C1V_REQUIRE(e == hipSuccess, "run_list_fork: join: %s", hipGetErrorString(e));
        open_side = false;
        return 0;
    };
    // An entry that fails ends the list; what was already issued on `side` is still joined, so that the caller's
    // synchronisation of `stream` covers it (the entry's own error text is kept).
    auto fail = [&]() {
        if (!open_side) return;
        const std::string keep = g_err;
        join();
        g_err = keep;
    };
    for (int32_t i = 0; i < n; ++i) C1V_REQUIRE(fns[i] != nullptr, "run_list_fork: entry %d has no function", i);
    for (int32_t i = 0; i < n; ++i) {
        const int f = flags[i];
        if (f & C1V_LIST_SIDE) {
            if (f & C1V_LIST_FORK) {
                hipError_t e = hipEventRecord(ev[dev][0], sm);
                if (e == hipSuccess) e = hipStreamWaitEvent(ss, ev[dev][0], 0);
                C1V_REQUIRE(e == hipSuccess, "run_list_fork: fork: %s", hipGetErrorString(e));
            }
            open_side = true;
            const int rc = fns[i](args[i], side);
            if (rc != 0) { fail(); return rc; }
        } else {
            if ((f & C1V_LIST_JOIN) && open_side && join()) return -1;
            const int rc = fns[i](args[i], stream);
            if (rc != 0) { fail(); return rc; }
        }
    }
    if (open_side && join()) return -1;
    return 0;
}
