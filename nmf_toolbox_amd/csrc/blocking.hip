// Blocking host-buffer entry points of the C ABI (what the MEX gateway binds): nmf / cnmf / lnmf / constrainednmf on one GPU or
// column-sharded over the GPUs of this process, ReconstructFromDecomposition, SortDictionary, projfunc.
#include <chrono>

#include "api_common.h"

using namespace nmfx;

namespace {

// per-source lambda / fixed flags (nmf.m:145-173 loops over the sources on one concatenated problem) as per-component vectors of length K >= K_total;
// components past K_total are the zero padding of the fused kernels: fixed, never updated
void expand_sources(const nmfx_problem *p, int K, std::vector<float> &lw, std::vector<float> &lh, std::vector<uint8_t> &fw, std::vector<uint8_t> &fh) {
    const int Kt = p->K_total;
    lw.assign(K, 0.f); lh.assign(K, 0.f); fw.assign(K, 0); fh.assign(K, 0);
    for (int k = Kt; k < K; ++k) fw[k] = fh[k] = 1;
    for (int s = 0, k0 = 0; s < p->num_sources; ++s) {
        const int Ks = p->K_s ? p->K_s[s] : Kt;
        for (int k = k0; k < k0 + Ks; ++k) {
            if (p->W_sparsity) lw[k] = (float)p->W_sparsity[s];
            if (p->H_sparsity) lh[k] = (float)p->H_sparsity[s];
            if (p->W_fixed) fw[k] = p->W_fixed[s];
            if (p->H_fixed) fh[k] = p->H_fixed[s];
        }
        k0 += Ks;
    }
}

// float64 host factors -> DEVICE doubles in the engine's (K-padded) layout, for nmfx_engine_init_f64: MATLAB's doubles reach the master copies unrounded.
// W: m x Kt x T -> m x K x T (zero columns appended to every time slice); H: columns [col0, col0 + ncols) of the Kt x n array -> K x ncols (zero rows appended)
nmfx_status stage_init64(hipStream_t st, const nmfx_problem *p, int K, long col0, long ncols, bool want_H, DevBuf &W0d, DevBuf &H0d) {
    const int Kt = p->K_total, T = p->T;
    const size_t sl = (size_t)p->m * Kt, slp = (size_t)p->m * K;
    TRY(W0d.alloc(slp * T * 8));
    if (K != Kt) NMFX_HIP(hipMemsetAsync(W0d.p, 0, slp * T * 8, st));
    for (int t = 0; t < T; ++t)
        NMFX_HIP(hipMemcpyAsync(W0d.as<double>() + t * slp, static_cast<const double *>(p->W_init) + t * sl, sl * 8, hipMemcpyHostToDevice, st));
    if (want_H) {
        const double *Hh = static_cast<const double *>(p->H_init) + (size_t)Kt * col0;
        TRY(H0d.alloc((size_t)K * ncols * 8));
        if (K != Kt) {
            DevBuf tmp;
            TRY(tmp.alloc((size_t)Kt * ncols * 8));
            NMFX_HIP(hipMemcpyAsync(tmp.p, Hh, (size_t)Kt * ncols * 8, hipMemcpyHostToDevice, st));
            TRY(repack_rows64(st, tmp.as<double>(), Kt, H0d.as<double>(), K, ncols));
            NMFX_HIP(hipStreamSynchronize(st));   // tmp goes out of scope
        } else NMFX_HIP(hipMemcpyAsync(H0d.p, Hh, (size_t)K * ncols * 8, hipMemcpyHostToDevice, st));
    }
    NMFX_HIP(hipStreamSynchronize(st));   // the host arrays are the caller's (pageable): the copies have left them
    return NMFX_OK;
}

nmfx_status run_mu(const nmfx_problem *p, nmfx_result *r, int algorithm, const int64_t *seg = nullptr, int64_t nz = 0, const void *Z_init = nullptr,
                   void *Z_out = nullptr) {
    TRY(validate_problem(p, r, false, algorithm != 3));
    if (algorithm != 1 && p->T != 1) { set_error("nmf / lnmf / constrainednmf: T must be 1"); return NMFX_ERR_INVALID; }
    if (algorithm == 3) {
        if (!seg || !Z_init || !Z_out || nz <= 0) { set_error("constrainednmf: segments, Z_init and Z_out are required"); return NMFX_ERR_INVALID; }
        if (p->num_sources != 1) { set_error("constrainednmf: single source only (constrainednmf.m has no multi-source form)"); return NMFX_ERR_INVALID; }
        if (p->divergence == NMFX_DIV_EUCLIDEAN_NOCOST) { set_error("constrainednmf: unknown divergence (constrainednmf.m:204-205)"); return NMFX_ERR_INVALID; }
    }
    if (algorithm == 0 && p->divergence == NMFX_DIV_EUCLIDEAN_NOCOST) { set_error("nmf: unknown divergence (nmf.m:165-166)"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(p->device));
    const int Kt = p->K_total;
    // K rounded up to a multiple of 32 with zero, fixed components opens the fused kernels to any K <= 256 on tileable shapes: the
    // padding contributes exact zeros to W*H and to every sum, and is never updated (it is stripped again on the way out)
    const int dv = p->divergence;
    // fused IS / alpha-beta (above K = 192, and the dual form alpha == 0, in two passes); constrainednmf has no dual-form kernels (fill_from_desc refuses
    // dualz for algorithm 3): padding K there would only widen the general path
    const bool dual_ok = (dv == NMFX_DIV_IS || dv == NMFX_DIV_AB) && Kt <= 256 && !(algorithm == 3 && dv == NMFX_DIV_AB && p->alpha == 0);
    int Kup = (Kt + 31) / 32 * 32;
    // cnmf: the same zero padding opens the fused shift-sum passes to any K below an instantiated (K, T) pair (K = 20, T = 8 runs as (32, 8); K = 20, T = 2 as
    // (64, 2), the smallest pair with that context length)
    if (algorithm == 1 && p->T > 1 && !fused_supported_T(Kup, p->T))
        for (int kk = Kup + 32; kk <= 256; kk += 32) if (fused_supported_T(kk, p->T)) { Kup = kk; break; }
    const bool pad_cnmf = algorithm == 1 && Kt != Kup && p->T > 1 && fused_supported_T(Kup, p->T) && p->m >= 64 && p->n >= 64 && p->path != 1 &&
                          (dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN || dv == NMFX_DIV_EUCLIDEAN_NOCOST ||
                           ((dv == NMFX_DIV_IS || (dv == NMFX_DIV_AB && p->alpha != 0)) && p->m % 4 == 0));   // (IS / alpha-beta: engine.fusedT_dual, every pair since round 6)
    const bool pad = pad_cnmf || (algorithm != 1 && Kt % 32 != 0 && (Kt <= 256 || ((dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN) && Kt <= 2048 && p->m >= 64 && p->n >= 64)) &&   // (above 256: column blocks, engine.klw / eucw)
                     ((p->m >= 64 && p->n >= 64) || p->path == 2) && p->path != 1 && (dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN || dual_ok));
    const int K = pad ? Kup : Kt;
    std::vector<float> lw, lh;
    std::vector<uint8_t> fw, fh;
    expand_sources(p, K, lw, lh, fw, fh);
    nmfx_engine_desc d{};
    d.m = p->m; d.n_local = p->n; d.K_total = K; d.T = p->T; d.divergence = p->divergence; d.alpha = p->alpha; d.beta = p->beta;
    d.lamW_col = lw.data(); d.lamH_row = lh.data(); d.fixW_col = fw.data(); d.fixH_row = fh.data();
    d.device = p->device; d.stream = nullptr; d.algorithm = algorithm; d.path = p->path;
    d.K_valid = pad ? Kt : 0;
    size_t ws_bytes = 0, packed_count = 0;
    TRY(nmfx_engine_workspace_bytes(&d, &ws_bytes));
    TRY(nmfx_engine_packed_count(&d, &packed_count));
    const size_t mn = (size_t)p->m * p->n, mKT = (size_t)p->m * K * p->T, Kn = (size_t)K * p->n;
    DevBuf V, W, H, Z, ws, packed, Wbak, dcost, tmp;   // (tmp: K x cols staging of the un-padded row-interleaved arrays H, Z)
    TRY(V.alloc(mn * 4)); TRY(W.alloc(mKT * 4)); TRY(H.alloc(Kn * 4)); TRY(packed.alloc(packed_count * 4));
    // everything else this call will ever allocate comes BEFORE the workspace: a workspace that only just fits must not starve them afterwards (the retry below
    // is for the workspace alone)
    const size_t mKt = (size_t)p->m * Kt * p->T, Ktn = (size_t)Kt * p->n;
    if (p->tolerance < 0) TRY(dcost.alloc(sizeof(double) * p->maxiter));
    else if (dv == NMFX_DIV_EUCLIDEAN || dv == NMFX_DIV_EUCLIDEAN_NOCOST) TRY(Wbak.alloc(mKT * 4));   // engines of cost lag 2 (known for sure only once the engine exists)
    if (pad) TRY(tmp.alloc(std::max(Ktn, (size_t)Kt * (size_t)(algorithm == 3 ? nz : 0)) * 4));
    if (algorithm == 3) TRY(Z.alloc((size_t)K * nz * 4));
    if (ws.alloc(ws_bytes) != NMFX_OK) {   // no room for the workspace with the transposed copy of V: the same problem without it (said in the descriptor, not guessed)
        (void)hipGetLastError();
        d.flags |= 1;
        TRY(nmfx_engine_workspace_bytes(&d, &ws_bytes));
        TRY(ws.alloc(ws_bytes));
    }
    hipStream_t st = nullptr;
    IoStats &io = io_stats();
    io = IoStats{};
    const auto t0 = std::chrono::steady_clock::now();
    TRY(upload(st, p->V, p->dtype, V.as<float>(), mn, 1.0));
    if (pad && p->T > 1) {   // cnmf: every time slice m x K of W is padded on its own
        const size_t sl = (size_t)p->m * Kt, slp = (size_t)p->m * K;
        NMFX_HIP(hipMemsetAsync(W.as<float>(), 0, mKT * 4, st));
        for (int t = 0; t < p->T; ++t) TRY(upload(st, static_cast<const char *>(p->W_init) + t * sl * dsize(p->dtype), p->dtype, W.as<float>() + t * slp, sl, 1.0));
    } else {
        TRY(upload(st, p->W_init, p->dtype, W.as<float>(), mKt, 1.0));   // the first K columns of the m x K_pad array
        if (pad) NMFX_HIP(hipMemsetAsync(W.as<float>() + mKt, 0, (mKT - mKt) * 4, st));
    }
    if (algorithm != 3) {
        if (pad) {
            TRY(upload(st, p->H_init, p->dtype, tmp.as<float>(), Ktn, 1.0));
            TRY(repack_rows(st, tmp.as<float>(), Kt, H.as<float>(), K, p->n));
        } else TRY(upload(st, p->H_init, p->dtype, H.as<float>(), Kn, 1.0));
    } else {   // H = Z*A is formed on the device by nmfx_engine_init (constrainednmf.m:174-177)
        if (pad) {
            TRY(upload(st, Z_init, p->dtype, tmp.as<float>(), (size_t)Kt * nz, 1.0));
            TRY(repack_rows(st, tmp.as<float>(), Kt, Z.as<float>(), K, nz));
        } else TRY(upload(st, Z_init, p->dtype, Z.as<float>(), (size_t)K * nz, 1.0));
    }
    // (declared after the buffers: on every return path the stream is drained and the engine destroyed BEFORE the buffers its kernels use are freed)
    struct EngineOwner {
        nmfx_engine *e = nullptr;
        hipStream_t st = nullptr;
        ~EngineOwner() { if (e) { (void)hipStreamSynchronize(st); nmfx_engine_destroy(e); } }
    } own;
    own.st = st;
    nmfx_engine *e = nullptr;
    TRY(nmfx_engine_create(&d, V.as<float>(), W.as<float>(), H.as<float>(), ws.p, ws_bytes, packed.as<float>(), &e));
    own.e = e;
    nmfx_status s = algorithm == 3 ? nmfx_engine_set_constraint(e, seg, nz, Z.as<float>()) : NMFX_OK;
    NMFX_HIP(hipStreamSynchronize(st));   // (nmfx_engine_create has drained the stream already: this only closes the ingest clock)
    const auto t1 = std::chrono::steady_clock::now();
    if (s == NMFX_OK && p->dtype == NMFX_F64) {   // float64 host buffers: the masters start from the caller's doubles
        DevBuf W0d, H0d;
        s = stage_init64(st, p, K, 0, p->n, algorithm != 3, W0d, H0d);
        if (s == NMFX_OK) s = nmfx_engine_init_f64(e, W0d.as<double>(), algorithm != 3 ? H0d.as<double>() : nullptr);
        if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();   // W0d / H0d go out of scope
    } else if (s == NMFX_OK) s = nmfx_engine_init(e);
    int it = 0;
    r->iters_run = 0;
    auto read_cost = [&](int idx) -> nmfx_status {
        hipError_t he = hipMemcpy(&r->cost[idx], e->cost, sizeof(double), hipMemcpyDeviceToHost);   // syncs the iteration
        if (he != hipSuccess) { set_error("cost readback: %s", hipGetErrorString(he)); return NMFX_ERR_HIP; }
        r->iters_run = idx + 1;
        return NMFX_OK;
    };
    // nmf.m:221-224 / cnmf.m:254-257
    auto stop = [&](int idx) {
        if (p->tolerance < 0 || idx == 0) return false;
        if (algorithm == 2) return r->cost[idx] <= r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] <= p->tolerance;   // lnmf.m:84
        return r->cost[idx] < r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] < p->tolerance;
    };
    bool stopped = false;
    const int lagk = e ? nmfx_engine_cost_lag(e) : 0;   // where cost(it-1) turns up: 1 after wstep_partial(it), 2 after wstep_finish(it), 0: cost(it) after hstep(it)
    const bool lag = lagk != 0;
    if (s == NMFX_OK && p->tolerance < 0) {
        // stop rule disabled (NMFX extension): nothing is decided on the host, so nothing is read back per iteration -- the costs land in a device
        // vector and come home once
        s = nmfx_engine_iterate(e, p->maxiter, dcost.as<double>());
        if (s == NMFX_OK && hipMemcpy(r->cost, dcost.p, sizeof(double) * p->maxiter, hipMemcpyDeviceToHost) != hipSuccess) { set_error("cost readback failed"); s = NMFX_ERR_HIP; }
        if (s == NMFX_OK) r->iters_run = p->maxiter;
        it = p->maxiter;
        stopped = true;   // (nothing left to finish below)
    }
    if (s == NMFX_OK && lagk == 2 && !stopped && !Wbak.p) s = Wbak.alloc(mKT * 4);
    for (it = stopped ? p->maxiter : 0; s == NMFX_OK && it < p->maxiter; ++it) {
        if ((s = nmfx_engine_wstep_partial(e)) != NMFX_OK) break;
        if (lagk == 1 && it > 0) {
            // the fused W-step pass of iteration it also yields cost(it-1); W and H are untouched until wstep_finish, so
            // stopping here returns exactly the state of iteration it-1 (the numerators just computed are discarded)
            if ((s = read_cost(it - 1)) != NMFX_OK) break;
            if (stop(it - 1)) { stopped = true; break; }
        }
        // Gram-form cost: cost(it-1) comes out of the W update itself, which has then already moved W -- keep the old W to hand back on a stop
        if (lagk == 2 && it > 0 && hipMemcpyAsync(Wbak.p, W.p, mKT * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("W backup failed"); s = NMFX_ERR_HIP; break; }
        if ((s = nmfx_engine_wstep_finish(e)) != NMFX_OK) break;
        if (lagk == 2 && it > 0) {
            if ((s = read_cost(it - 1)) != NMFX_OK) break;
            if (stop(it - 1)) {
                if (hipMemcpyAsync(W.p, Wbak.p, mKT * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("W restore failed"); s = NMFX_ERR_HIP; }
                stopped = true;
                break;
            }
        }
        if ((s = nmfx_engine_hstep(e)) != NMFX_OK) break;
        if (!lag) {
            if ((s = read_cost(it)) != NMFX_OK) break;
            if (stop(it)) { stopped = true; break; }
        }
    }
    if (s == NMFX_OK && lag && !stopped) {
        s = nmfx_engine_cost_pass(e);
        if (s == NMFX_OK) s = read_cost(p->maxiter - 1);
    }
    r->cost_len = r->iters_run;
    if (algorithm == 2) {   // lnmf.m:84-86 breaks WITHOUT trimming: the cost vector keeps its maxiter length, zero after the stop
        for (int i = r->iters_run; i < p->maxiter; ++i) r->cost[i] = 0.0;
        r->cost_len = p->maxiter;
    }
    const auto t2 = std::chrono::steady_clock::now();   // (the last cost read-back has synchronised the iterations)
    if (pad && p->T > 1) {
        const size_t sl = (size_t)p->m * Kt, slp = (size_t)p->m * K;
        for (int t = 0; t < p->T && s == NMFX_OK; ++t) s = download(st, W.as<float>() + t * slp, p->dtype, static_cast<char *>(r->W) + t * sl * dsize(p->dtype), sl);
    } else if (s == NMFX_OK) s = download(st, W.as<float>(), p->dtype, r->W, mKt);
    if (s == NMFX_OK && pad) {
        s = repack_rows(st, H.as<float>(), K, tmp.as<float>(), Kt, p->n);
        if (s == NMFX_OK) s = download(st, tmp.as<float>(), p->dtype, r->H, Ktn);
        if (s == NMFX_OK && algorithm == 3) s = repack_rows(st, Z.as<float>(), K, tmp.as<float>(), Kt, nz);
        if (s == NMFX_OK && algorithm == 3) s = download(st, tmp.as<float>(), p->dtype, Z_out, (size_t)Kt * nz);
    } else {
        if (s == NMFX_OK) s = download(st, H.as<float>(), p->dtype, r->H, Kn);
        if (s == NMFX_OK && algorithm == 3) s = download(st, Z.as<float>(), p->dtype, Z_out, (size_t)K * nz);
    }
    (void)hipStreamSynchronize(st);
    nmfx_engine_destroy(e);
    own.e = nullptr;
    const auto t3 = std::chrono::steady_clock::now();
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    io.ingest_s = sec(t0, t1); io.iterate_s = sec(t1, t2); io.egress_s = sec(t2, t3);
    return s;
}

// ---- nmfx_problem.n_gpus > 1: one process, one host thread, one stream + engine per device (what a MEX caller of nmf() needs) -------
// V and H are column-sharded over the devices, W is replicated.  Per iteration ONE exchange of the packed W-step sums
// (SURVEY 8(e)), done here without a collective library: every device reduces its own 1/N slice of `packed` straight out of its
// peers' HBM over xGMI (all links in parallel, fixed summation order), then copies the other N-1 reduced slices from their owners.
// Each slice has exactly one owner, so all replicas of W stay bit-identical.  device_ids may name one device several times
// (N shards on one GPU): that is how the 1-GPU test box exercises this path.
struct MultiDev {
    int ndev = 0;
    int dev[NMFX_MAX_GPUS];
    hipStream_t st[NMFX_MAX_GPUS] = {};
    hipEvent_t evP[NMFX_MAX_GPUS] = {}, evR[NMFX_MAX_GPUS] = {}, evG[NMFX_MAX_GPUS] = {}, evH[NMFX_MAX_GPUS] = {};
    nmfx_engine *eng[NMFX_MAX_GPUS] = {};
    DevBuf V[NMFX_MAX_GPUS], W[NMFX_MAX_GPUS], H[NMFX_MAX_GPUS], ws[NMFX_MAX_GPUS], packed[NMFX_MAX_GPUS], costh[NMFX_MAX_GPUS], tmp[NMFX_MAX_GPUS];
    long lo[NMFX_MAX_GPUS + 1];
    long hL[NMFX_MAX_GPUS] = {}, hR[NMFX_MAX_GPUS] = {};   // cnmf: T-1 halo columns of H on each inner edge (and of V on the right one)
    // host side of the small device <-> host scalars (per-shard cost partials, ||V||^2): pinned.  They used to be async copies into a std::vector / the
    // stack; a rare host-heap corruption ("free(): invalid pointer", scripts/fuzz_campaign_r3.py multi_edge, only with the NumPy oracle's threads alive in
    // the same process) went away with them -- asynchronous copies into a few bytes of pageable heap are staged by the runtime
    double *hpin = nullptr;
    bool use_rccl = false;                 // the packed exchange: RCCL (rccl_backend.hip) or the peer reduce-scatter + all-gather below
    void *comms[NMFX_MAX_GPUS] = {};
    int lease_n = 0;                       // > 0: this call holds the cached RCCL communicator set of dev[0 .. lease_n) (rccl_comms) and hands it back below
    std::vector<hipEvent_t> evX;           // pairs around the first exchanges on device 0's stream (nmfx_last_call_exchange)
    int nx = 0;
    nmfx_status init_host() {
        if (!hpin) NMFX_HIP(hipHostMalloc(reinterpret_cast<void **>(&hpin), sizeof(double) * (NMFX_MAX_GPUS + 2), hipHostMallocPortable));
        return NMFX_OK;
    }
    ~MultiDev() {
        for (int g = 0; g < ndev; ++g) {   // an error path may leave work in flight that reads the peers' buffers: drain every stream before anything is freed
            (void)hipSetDevice(dev[g]);
            if (st[g]) (void)hipStreamSynchronize(st[g]);
        }
        staging_quiesce();   // (the ingest left its DMA-done events recorded on these streams)
        if (lease_n > 0) rccl_release(dev, lease_n);   // (every collective of this call has completed: the streams are drained)
        for (int g = 0; g < ndev; ++g) {
            (void)hipSetDevice(dev[g]);
            if (eng[g]) nmfx_engine_destroy(eng[g]);
            unpool_event(dev[g], evP[g]); unpool_event(dev[g], evR[g]); unpool_event(dev[g], evG[g]); unpool_event(dev[g], evH[g]);
            unpool_stream(dev[g], st[g]);   // (drained above)
        }
        if (ndev > 0) (void)hipSetDevice(dev[0]);
        for (hipEvent_t ev : evX) unpool_event_timed(dev[0], ev);
        if (hpin) (void)hipHostFree(hpin);
    }
};

nmfx_status multi_allreduce_peer(MultiDev &M, size_t count);
// the ONE exchange of an iteration: packed[g] <- sum over the devices, in place, on every device's own stream
nmfx_status multi_allreduce(MultiDev &M, size_t count) {
    const bool timed = M.nx < 32;
    if (timed) {
        NMFX_HIP(hipSetDevice(M.dev[0]));
        hipEvent_t a = nullptr, b = nullptr;
        TRY(pool_event_timed(M.dev[0], &a)); TRY(pool_event_timed(M.dev[0], &b));
        M.evX.push_back(a); M.evX.push_back(b);
        NMFX_HIP(hipEventRecord(a, M.st[0]));
    }
    if (M.use_rccl) {
        float *bufs[NMFX_MAX_GPUS];
        for (int g = 0; g < M.ndev; ++g) bufs[g] = M.packed[g].as<float>();
        TRY(rccl_allreduce_f32(M.comms, M.dev, M.st, bufs, M.ndev, count));
    } else TRY(multi_allreduce_peer(M, count));
    if (timed) {
        NMFX_HIP(hipSetDevice(M.dev[0]));
        NMFX_HIP(hipEventRecord(M.evX[2 * M.nx + 1], M.st[0]));
        ++M.nx;
    }
    return NMFX_OK;
}
nmfx_status multi_allreduce_peer(MultiDev &M, size_t count) {
    const int N = M.ndev;
    PeerPtrs ptrs{};
    for (int g = 0; g < N; ++g) ptrs.p[g] = M.packed[g].as<float>();
    const long per = (long)(((count + N - 1) / N + 3) & ~(size_t)3);   // slice length, a multiple of 4 floats
    auto slice = [&](int g, long *off, long *cnt) { *off = std::min((long)count, per * g); *cnt = std::min((long)count, per * (g + 1)) - *off; };
    for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipEventRecord(M.evP[g], M.st[g])); }
    for (int g = 0; g < N; ++g) {   // reduce-scatter: device g owns slice g
        NMFX_HIP(hipSetDevice(M.dev[g]));
        for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evP[h], 0));
        long off, cnt;
        slice(g, &off, &cnt);
        TRY(peer_reduce(M.st[g], ptrs, N, g, off, cnt));
        NMFX_HIP(hipEventRecord(M.evR[g], M.st[g]));
    }
    for (int g = 0; g < N; ++g) {   // all-gather: fetch the slices the others reduced
        NMFX_HIP(hipSetDevice(M.dev[g]));
        for (int h = 0; h < N; ++h) {
            if (h == g) continue;
            long off, cnt;
            slice(h, &off, &cnt);
            NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evR[h], 0));
            if (cnt > 0) NMFX_HIP(hipMemcpyPeerAsync(ptrs.p[g] + off, M.dev[g], ptrs.p[h] + off, M.dev[h], (size_t)cnt * 4, M.st[g]));
        }
        NMFX_HIP(hipEventRecord(M.evG[g], M.st[g]));
    }
    for (int g = 0; g < N; ++g) {   // nobody refills its `packed` (next W-step partial) before every peer has copied out of it
        NMFX_HIP(hipSetDevice(M.dev[g]));
        for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evG[h], 0));
    }
    return NMFX_OK;
}

// cnmf on column shards (cnmf.m:188,219 shift across the shard edges): after an H update every device fetches the T-1 columns next to
// each of its inner edges from the neighbour that owns them, point-to-point over xGMI, on its own stream behind the neighbour's update
nmfx_status multi_halo_exchange(MultiDev &M, int K, int hh) {
    const int N = M.ndev;
    for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipEventRecord(M.evH[g], M.st[g])); }
    const size_t bytes = (size_t)K * hh * 4;
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(M.dev[g]));
        const long nl = M.lo[g + 1] - M.lo[g];
        if (g > 0) {   // my left halo = the last T-1 columns of the left neighbour
            const long nln = M.lo[g] - M.lo[g - 1];
            NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evH[g - 1], 0));
            NMFX_HIP(hipMemcpyPeerAsync(M.H[g].as<float>(), M.dev[g], M.H[g - 1].as<float>() + (size_t)K * (M.hL[g - 1] + nln - hh), M.dev[g - 1], bytes, M.st[g]));
        }
        if (g < N - 1) {   // my right halo = the first T-1 columns of the right neighbour
            NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evH[g + 1], 0));
            NMFX_HIP(hipMemcpyPeerAsync(M.H[g].as<float>() + (size_t)K * (M.hL[g] + nl), M.dev[g], M.H[g + 1].as<float>() + (size_t)K * M.hL[g + 1], M.dev[g + 1], bytes, M.st[g]));
        }
    }
    // (the next H update of a neighbour comes after the next packed exchange, which waits for every device's stream: no copy is still reading then)
    return NMFX_OK;
}

nmfx_status run_mu_multi(const nmfx_problem *p, nmfx_result *r, int algorithm) {
    TRY(validate_problem(p, r, false, true));
    if (algorithm != 0 && algorithm != 1 && algorithm != 2) { set_error("n_gpus > 1 is implemented for nmf, cnmf, lnmf and nmfsc"); return NMFX_ERR_UNSUPPORTED; }
    if (algorithm != 1 && p->T != 1) { set_error("nmf / lnmf: T must be 1"); return NMFX_ERR_INVALID; }
    if (algorithm == 0 && p->divergence == NMFX_DIV_EUCLIDEAN_NOCOST) { set_error("nmf: unknown divergence (nmf.m:165-166)"); return NMFX_ERR_INVALID; }
    const int N = p->n_gpus;
    const int T = p->T, hh = T - 1;
    if (N > NMFX_MAX_GPUS || N > p->n) { set_error("n_gpus = %d: at most %d devices and one column per device", N, NMFX_MAX_GPUS); return NMFX_ERR_INVALID; }
    if (hh > 0 && p->n / N < hh) { set_error("cnmf on %d devices: every shard needs at least T-1 = %d columns", N, hh); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    MultiDev M;
    TRY(M.init_host());
    for (int g = 0; g < N; ++g) {
        M.dev[g] = p->device_ids ? p->device_ids[g] : g;
        TRY(check_device(M.dev[g]));
    }
    {   // which exchange: nmfx_problem.multi_backend, NMFX_MULTI_BACKEND for "auto"
        int want = p->multi_backend;
        if (want == 0) { const char *env = getenv("NMFX_MULTI_BACKEND"); if (env) want = !strcmp(env, "rccl") ? 2 : (!strcmp(env, "peer") ? 1 : 0); }
        std::string why;
        if (want == 2) {
            if (!rccl_usable(M.dev, N, &why)) { set_error("multi_backend = rccl: %s", why.c_str()); return NMFX_ERR_UNSUPPORTED; }
            M.use_rccl = true;
        } else M.use_rccl = want == 0 && rccl_usable(M.dev, N, &why);
        if (M.use_rccl) { TRY(rccl_comms(M.dev, N, M.comms)); M.lease_n = N; }
    }
    for (int g = 0; g < N; ++g)      // peer mappings: the reduce kernel reads the other devices' `packed` in place (and cnmf's halo copies go direct)
        for (int h = 0; h < N; ++h) {
            if (M.dev[g] == M.dev[h]) continue;
            int can = 0;
            NMFX_HIP(hipDeviceCanAccessPeer(&can, M.dev[g], M.dev[h]));
            if (!can) { set_error("device %d cannot access device %d as a peer", M.dev[g], M.dev[h]); return NMFX_ERR_UNSUPPORTED; }
            NMFX_HIP(hipSetDevice(M.dev[g]));
            hipError_t pe = hipDeviceEnablePeerAccess(M.dev[h], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", M.dev[g], M.dev[h], hipGetErrorString(pe)); return NMFX_ERR_HIP; }
            (void)hipGetLastError();
        }
    const int Kt = p->K_total, dv = p->divergence;
    const long m = p->m, n = p->n;
    M.lo[0] = 0;
    for (int g = 0; g < N; ++g) M.lo[g + 1] = M.lo[g] + n / N + (g < n % N ? 1 : 0);   // contiguous column blocks, as engine.shard_columns
    long nmin = n;
    for (int g = 0; g < N; ++g) nmin = std::min(nmin, M.lo[g + 1] - M.lo[g]);
    const bool dual_ok = (dv == NMFX_DIV_IS || dv == NMFX_DIV_AB) && Kt <= 256 && !(algorithm == 3 && dv == NMFX_DIV_AB && p->alpha == 0);
    const bool pad = algorithm != 1 && Kt % 32 != 0 && (Kt <= 256 || ((dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN) && Kt <= 2048 && m >= 64 && nmin >= 64)) && ((m >= 64 && nmin >= 64) || p->path == 2) &&
                     p->path != 1 && (dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN || dual_ok);
    const int K = pad ? (Kt + 31) / 32 * 32 : Kt;
    std::vector<float> lw, lh;
    std::vector<uint8_t> fw, fh;
    expand_sources(p, K, lw, lh, fw, fh);
    // every shard must run the same kernels (the packed layout and the summation order of the replicated W update depend on them): the fused paths want at
    // least 64 columns, so one short shard sends all of them to the general kernels
    const int shard_path = (p->path == 0 && nmin < 64) ? 1 : p->path;
    const size_t mK = (size_t)m * K * T, mKt = (size_t)m * Kt * T;   // (cnmf: the T slices of W; K is never padded there)
    size_t packed_count = 0, wsb[NMFX_MAX_GPUS] = {};
    int kind = -1;
    nmfx_engine_desc dd[NMFX_MAX_GPUS];
    DevBuf Wbak;   // Gram-form cost + stop rule: device 0's W as it was before the update that produced cost(it-1)
    // pass 1: streams, events and every buffer except the workspaces.  The workspaces come last and ALL at once, because whether they hold the transposed copy
    // of V is one decision for the whole call (the kernel path, and with it the summation order of the replicated W update, follows from the descriptor)
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(M.dev[g]));
        M.ndev = g + 1;
        TRY(pool_stream(M.dev[g], &M.st[g]));
        TRY(pool_event(M.dev[g], &M.evP[g])); TRY(pool_event(M.dev[g], &M.evR[g])); TRY(pool_event(M.dev[g], &M.evG[g])); TRY(pool_event(M.dev[g], &M.evH[g]));
        const long nl = M.lo[g + 1] - M.lo[g];
        const long hL = M.hL[g] = g > 0 ? hh : 0, hR = M.hR[g] = g < N - 1 ? hh : 0;   // H = [left halo | own columns | right halo], V = [own | right halo]
        nmfx_engine_desc &d = dd[g];
        d = nmfx_engine_desc{};
        d.m = m; d.n_local = nl; d.K_total = K; d.T = T; d.divergence = dv; d.alpha = p->alpha; d.beta = p->beta;
        d.halo_left = (int)hL; d.halo_right = (int)hR; d.n_valid = nl + hR;
        d.lamW_col = lw.data(); d.lamH_row = lh.data(); d.fixW_col = fw.data(); d.fixH_row = fh.data();
        d.device = M.dev[g]; d.stream = M.st[g]; d.algorithm = algorithm; d.path = shard_path; d.K_valid = pad ? Kt : 0; d.col_offset = M.lo[g];
        size_t pc = 0;
        TRY(nmfx_engine_packed_count(&d, &pc));
        if (g == 0) packed_count = pc;
        else if (pc != packed_count) { set_error("n_gpus: shards disagree on the packed layout"); return NMFX_ERR_INVALID; }
        const long nh = hL + nl + hR;
        TRY(M.V[g].alloc((size_t)m * (nl + hR) * 4)); TRY(M.W[g].alloc(mK * 4)); TRY(M.H[g].alloc((size_t)K * nh * 4));
        TRY(M.packed[g].alloc(pc * 4)); TRY(M.costh[g].alloc(64));
        if (pad) TRY(M.tmp[g].alloc((size_t)Kt * nl * 4));
        if (g == 0 && p->tolerance >= 0 && (dv == NMFX_DIV_EUCLIDEAN || dv == NMFX_DIV_EUCLIDEAN_NOCOST)) TRY(Wbak.alloc(mK * 4));
    }
    // pass 2: the workspaces, with the transposed copy of V; if ONE of them does not fit, every shard runs without it (flags bit 0 on all of them)
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool ok = true;
        for (int g = 0; g < N && ok; ++g) {
            NMFX_HIP(hipSetDevice(M.dev[g]));
            dd[g].flags = attempt == 0 ? 0 : 1;
            TRY(nmfx_engine_workspace_bytes(&dd[g], &wsb[g]));
            if (M.ws[g].alloc(wsb[g]) != NMFX_OK) {
                if (attempt == 1) return NMFX_ERR_NOMEM;   // (the message of the failed allocation stands)
                (void)hipGetLastError();
                ok = false;
            }
        }
        if (ok) break;
        for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); M.ws[g].release(); }
    }
    // pass 3: ingest and engines.  The clocks of nmfx_last_call_timing belong to THIS call (run_mu stamps the same three spans): ingest = host arrays in +
    // engines + init, iterate = the loop incl. the closing cost pass, egress = results out
    IoStats &io = io_stats();
    io = IoStats{};
    const auto t0 = std::chrono::steady_clock::now();
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(M.dev[g]));
        const long nl = M.lo[g + 1] - M.lo[g], hL = M.hL[g], hR = M.hR[g], nh = hL + nl + hR;
        const char *Vh = static_cast<const char *>(p->V) + (size_t)m * M.lo[g] * dsize(p->dtype);           // a column block is a contiguous slab
        const char *Hh = static_cast<const char *>(p->H_init) + (size_t)Kt * (M.lo[g] - hL) * dsize(p->dtype);
        TRY(upload(M.st[g], Vh, p->dtype, M.V[g].as<float>(), (size_t)m * (nl + hR), 1.0));
        TRY(upload(M.st[g], p->W_init, p->dtype, M.W[g].as<float>(), mKt, 1.0));
        if (pad) {
            NMFX_HIP(hipMemsetAsync(M.W[g].as<float>() + mKt, 0, (mK - mKt) * 4, M.st[g]));
            TRY(upload(M.st[g], Hh, p->dtype, M.tmp[g].as<float>(), (size_t)Kt * nl, 1.0));
            TRY(repack_rows(M.st[g], M.tmp[g].as<float>(), Kt, M.H[g].as<float>(), K, nl));
        } else TRY(upload(M.st[g], Hh, p->dtype, M.H[g].as<float>(), (size_t)K * nh, 1.0));
        TRY(nmfx_engine_create(&dd[g], M.V[g].as<float>(), M.W[g].as<float>(), M.H[g].as<float>(), M.ws[g].p, wsb[g], M.packed[g].as<float>(), &M.eng[g]));
        TRY(nmfx_engine_set_rank0(M.eng[g], g == 0));
        if (hL || hR) TRY(nmfx_engine_defer_hstep_finish(M.eng[g], 1));   // V_hat / cost only once the neighbours' new columns are in
        const int kd = nmfx_engine_is_fused(M.eng[g]);
        if (kind < 0) kind = kd;
        else if (kd != kind) { set_error("n_gpus: shards picked different kernel paths; pass path = 1"); return NMFX_ERR_UNSUPPORTED; }
        if (p->dtype == NMFX_F64) {   // float64 host buffers: the masters start from the caller's doubles (this shard's own columns of H)
            DevBuf W0d, H0d;
            TRY(stage_init64(M.st[g], p, K, M.lo[g], nl, true, W0d, H0d));
            nmfx_status si = nmfx_engine_init_f64(M.eng[g], W0d.as<double>(), H0d.as<double>());
            NMFX_HIP(hipStreamSynchronize(M.st[g]));   // W0d / H0d go out of scope
            TRY(si);
        } else TRY(nmfx_engine_init(M.eng[g]));
    }
    if (hh > 0) {   // the halo columns were scaled as fp32 copies (cnmf.m:165): fetch the owners' images instead, so that every shard sees the same H
        TRY(multi_halo_exchange(M, K, hh));
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_hstep_finish(M.eng[g]));   // (paths that keep V_hat: refreshed with the final halos)
        for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipStreamSynchronize(M.st[g])); }
    }
    const int lagk = nmfx_engine_cost_lag(M.eng[0]);   // 1: cost(it-1) after wstep_partial(it); 2: after wstep_finish(it) (Gram-form cost); 0: cost(it) after hstep(it)
    const bool lag = lagk != 0;
    {   // Gram-form cost: every shard's mode decision uses the GLOBAL ||V||^2
        double vv = 0.0;
        double &part = M.hpin[NMFX_MAX_GPUS], &vvp = M.hpin[NMFX_MAX_GPUS + 1];
        for (int g = 0; g < N; ++g) {
            NMFX_HIP(hipSetDevice(M.dev[g]));
            TRY(nmfx_engine_sumvv_local(M.eng[g], M.costh[g].as<double>()));
            NMFX_HIP(hipMemcpyAsync(&part, M.costh[g].p, sizeof(double), hipMemcpyDeviceToHost, M.st[g]));
            NMFX_HIP(hipStreamSynchronize(M.st[g]));
            vv += part;
        }
        for (int g = 0; g < N; ++g) {
            NMFX_HIP(hipSetDevice(M.dev[g]));
            vvp = vv;
            NMFX_HIP(hipMemcpyAsync(M.costh[g].p, &vvp, sizeof(double), hipMemcpyHostToDevice, M.st[g]));
            TRY(nmfx_engine_sumvv_set_global(M.eng[g], M.costh[g].as<double>()));
            NMFX_HIP(hipStreamSynchronize(M.st[g]));   // vv is a stack variable
        }
    }
    if (lagk == 2 && p->tolerance >= 0 && !Wbak.p) { NMFX_HIP(hipSetDevice(M.dev[0])); TRY(Wbak.alloc(mK * 4)); }
    for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipStreamSynchronize(M.st[g])); }   // closes the ingest clock (init queued its kernels)
    const auto t1 = std::chrono::steady_clock::now();
    double *hc = M.hpin;   // pinned: the 8-byte read-backs land by DMA, not through the runtime's staging of pageable memory (see MultiDev::hpin)
    auto read_cost = [&](int idx) -> nmfx_status {   // cost = sum of the shards' partials (the lambda*|W| term lives on device 0 only)
        for (int g = 0; g < N; ++g) {
            NMFX_HIP(hipSetDevice(M.dev[g]));
            NMFX_HIP(hipMemcpyAsync(&hc[g], M.eng[g]->cost, sizeof(double), hipMemcpyDeviceToHost, M.st[g]));
        }
        double c = 0.0;
        for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipStreamSynchronize(M.st[g])); c += hc[g]; }
        r->cost[idx] = c;
        r->iters_run = idx + 1;
        return NMFX_OK;
    };
    auto stop = [&](int idx) {
        if (p->tolerance < 0 || idx == 0) return false;
        if (algorithm == 2) return r->cost[idx] <= r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] <= p->tolerance;   // lnmf.m:84
        return r->cost[idx] < r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] < p->tolerance;                         // nmf.m:221
    };
    r->iters_run = 0;
    bool stopped = false;
    for (int it = 0; it < p->maxiter; ++it) {
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_wstep_partial(M.eng[g]));
        if (lagk == 1 && it > 0) {
            TRY(read_cost(it - 1));
            if (stop(it - 1)) { stopped = true; break; }
        }
        TRY(multi_allreduce(M, packed_count));
        if (lagk == 2 && it > 0 && Wbak.p) { NMFX_HIP(hipSetDevice(M.dev[0])); NMFX_HIP(hipMemcpyAsync(Wbak.p, M.W[0].p, mK * 4, hipMemcpyDeviceToDevice, M.st[0])); }
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_wstep_finish(M.eng[g]));
        if (lagk == 2 && it > 0 && p->tolerance >= 0) {
            TRY(read_cost(it - 1));
            if (stop(it - 1)) {   // the W that is handed back (device 0's replica) as it was when iteration it-1 ended; H has not moved yet
                NMFX_HIP(hipSetDevice(M.dev[0]));
                NMFX_HIP(hipMemcpyAsync(M.W[0].p, Wbak.p, mK * 4, hipMemcpyDeviceToDevice, M.st[0]));
                stopped = true;
                break;
            }
        } else if (lagk == 2 && it > 0) TRY(read_cost(it - 1));
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_hstep(M.eng[g]));
        if (hh > 0) {
            TRY(multi_halo_exchange(M, K, hh));
            for (int g = 0; g < N; ++g) TRY(nmfx_engine_hstep_finish(M.eng[g]));
        }
        if (!lag) {
            TRY(read_cost(it));
            if (stop(it)) { stopped = true; break; }
        }
    }
    if (lag && !stopped) {
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_cost_pass(M.eng[g]));
        TRY(read_cost(p->maxiter - 1));
    }
    r->cost_len = r->iters_run;
    if (algorithm == 2) {
        for (int i = r->iters_run; i < p->maxiter; ++i) r->cost[i] = 0.0;
        r->cost_len = p->maxiter;
    }
    for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipStreamSynchronize(M.st[g])); }   // (a stop leaves the speculative W-step partials of the other shards in flight)
    const auto t2 = std::chrono::steady_clock::now();
    {   // the exchange as device 0's stream saw it
        io.exchange_backend = M.use_rccl ? 2 : 1; io.exchange_ms = 0; io.exchanges_timed = 0;
        NMFX_HIP(hipSetDevice(M.dev[0]));
        NMFX_HIP(hipStreamSynchronize(M.st[0]));
        for (int i = 0; i < M.nx; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, M.evX[2 * i], M.evX[2 * i + 1]) == hipSuccess) { io.exchange_ms += ms; io.exchanges_timed++; } else (void)hipGetLastError();
        }
    }
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(M.dev[g]));
        const long nl = M.lo[g + 1] - M.lo[g];
        if (g == 0) TRY(download(M.st[g], M.W[g].as<float>(), p->dtype, r->W, mKt));
        char *Hh = static_cast<char *>(r->H) + (size_t)Kt * M.lo[g] * dsize(p->dtype);
        if (pad) {
            TRY(repack_rows(M.st[g], M.H[g].as<float>(), K, M.tmp[g].as<float>(), Kt, nl));
            TRY(download(M.st[g], M.tmp[g].as<float>(), p->dtype, Hh, (size_t)Kt * nl));
        } else TRY(download(M.st[g], M.H[g].as<float>() + (size_t)K * M.hL[g], p->dtype, Hh, (size_t)K * nl));
    }
    for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipStreamSynchronize(M.st[g])); }
    const auto t3 = std::chrono::steady_clock::now();
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    io.ingest_s = sec(t0, t1); io.iterate_s = sec(t1, t2); io.egress_s = sec(t2, t3);
    return NMFX_OK;
}

}  // namespace

extern "C" {

// n_gpus / device_ids: a one-entry list names THE device (it overrides p->device); more entries shard the columns
static nmfx_status dispatch_mu(const nmfx_problem *p, nmfx_result *r, int algorithm) {
    if (!p) return run_mu(p, r, algorithm);
    if (p->n_gpus > 1 || (p->n_gpus == 1 && p->multi_backend != 0)) return run_mu_multi(p, r, algorithm);   // (one shard through the sharded driver: how a 1-GPU box runs the RCCL branch)
    if (p->n_gpus == 1 && p->device_ids) { nmfx_problem q = *p; q.device = p->device_ids[0]; return run_mu(&q, r, algorithm); }
    return run_mu(p, r, algorithm);
}
nmfx_status nmfx_nmf(const nmfx_problem *p, nmfx_result *r) { return dispatch_mu(p, r, 0); }
nmfx_status nmfx_cnmf(const nmfx_problem *p, nmfx_result *r) { return dispatch_mu(p, r, 1); }
nmfx_status nmfx_lnmf(const nmfx_problem *p, nmfx_result *r) { return dispatch_mu(p, r, 2); }
nmfx_status nmfx_constrainednmf(const nmfx_problem *p, const int64_t *segments, int64_t nz, const void *Z_init, nmfx_result *r, void *Z_out) {
    return run_mu(p, r, 3, segments, nz, Z_init, Z_out);
}

nmfx_status nmfx_reconstruct(int64_t m, int64_t n, int32_t K, int32_t T, int32_t dtype, const void *W, const void *H, void *V_hat,
                             int32_t device) {
    if (m <= 0 || n <= 0 || K <= 0 || T <= 0 || !W || !H || !V_hat) { set_error("nmfx_reconstruct: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(device));
    const size_t mn = (size_t)m * n, mKT = (size_t)m * K * T, Kn = (size_t)K * n;
    DevBuf Wd, Hd, Vd;
    TRY(Wd.alloc(mKT * 4)); TRY(Hd.alloc(Kn * 4)); TRY(Vd.alloc(mn * 4));
    hipStream_t st = nullptr;
    StreamDrain drain_(st);
    TRY(upload(st, W, dtype, Wd.as<float>(), mKT, 1.0));
    TRY(upload(st, H, dtype, Hd.as<float>(), Kn, 1.0));
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = m; g.N = n; g.Kc = (long)K * T;
    g.A = OpView{Wd.as<float>(), nullptr, (long)m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    if (T == 1) g.B = OpView{Hd.as<float>(), nullptr, (long)K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    else g.B = OpView{Hd.as<float>(), nullptr, (long)K, VIEW_HSTACK_KC, K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    g.C = Vd.as<float>(); g.ldc = m; g.epi = EPI_STORE; g.splitk = 1;
    TRY(launch_gemm(st, g));
    return download(st, Vd.as<float>(), dtype, V_hat, mn);
}

// [W_sorted, H_sorted] = SortDictionary(W, H): basis columns by increasing centre of mass (SortDictionary.m:33-47), computed in the
// buffers' own dtype; H / H_sorted may be NULL (nargin < 2).  order_out[K] receives the 0-based permutation (`sorted` - 1).
nmfx_status nmfx_sort_dictionary(int64_t m, int32_t K, int64_t n, int32_t dtype, const void *W, const void *H, void *W_sorted, void *H_sorted,
                                 int32_t *order_out, int32_t device) {
    if (m <= 0 || K <= 0 || !W || !W_sorted || (H && (!H_sorted || n <= 0))) { set_error("nmfx_sort_dictionary: bad arguments"); return NMFX_ERR_INVALID; }
    if (dtype != NMFX_F32 && dtype != NMFX_F64) { set_error("dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(device));
    const size_t es = dsize(dtype), wb = (size_t)m * K * es, hb = H ? (size_t)K * n * es : 0;
    DevBuf Wd, Ws, Hd, Hs, cog, ord;
    TRY(Wd.alloc(wb)); TRY(Ws.alloc(wb)); TRY(cog.alloc(sizeof(int) * K)); TRY(ord.alloc(sizeof(int) * K));
    hipStream_t st = nullptr;
    std::vector<int> cg(K), order(K);
    StreamDrain drain_(st);   // (after the vectors and the buffers: drained before they die on any return path)
    NMFX_HIP(hipMemcpyAsync(Wd.p, W, wb, hipMemcpyHostToDevice, st));
    TRY(center_of_gravity(st, Wd.p, dtype == NMFX_F64, m, K, cog.as<int>()));
    NMFX_HIP(hipMemcpyAsync(cg.data(), cog.p, sizeof(int) * K, hipMemcpyDeviceToHost, st));
    NMFX_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < K; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cg[a] < cg[b]; });   // MATLAB sort is stable (SortDictionary.m:43)
    NMFX_HIP(hipMemcpyAsync(ord.p, order.data(), sizeof(int) * K, hipMemcpyHostToDevice, st));
    TRY(permute(st, Wd.p, Ws.p, dtype == NMFX_F64, m, K, ord.as<int>(), 0));
    NMFX_HIP(hipMemcpyAsync(W_sorted, Ws.p, wb, hipMemcpyDeviceToHost, st));
    if (H) {
        nmfx_status sa = Hd.alloc(hb);
        if (sa == NMFX_OK) sa = Hs.alloc(hb);
        if (sa != NMFX_OK) return sa;
        NMFX_HIP(hipMemcpyAsync(Hd.p, H, hb, hipMemcpyHostToDevice, st));
        TRY(permute(st, Hd.p, Hs.p, dtype == NMFX_F64, K, n, ord.as<int>(), 1));
        NMFX_HIP(hipMemcpyAsync(H_sorted, Hs.p, hb, hipMemcpyDeviceToHost, st));
    }
    NMFX_HIP(hipStreamSynchronize(st));
    if (order_out) for (int k = 0; k < K; ++k) order_out[k] = order[k];
    return NMFX_OK;
}

nmfx_status nmfx_projfunc_dev(void *stream, float *X_dev, int64_t N, int32_t count, double k1, double k2, int32_t nn, const float *src_dev,
                              const float *dir_dev, double mu, int32_t *usediters_dev) {
    if (N <= 0 || count <= 0 || !X_dev) { set_error("nmfx_projfunc_dev: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;   // launch on the device the vectors live on, whatever the caller's current device is
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, X_dev) != hipSuccess) { (void)hipGetLastError(); set_error("nmfx_projfunc_dev: X_dev is not a device pointer"); return NMFX_ERR_INVALID; }
    TRY(check_device(attr.device));
    return projfunc_cols(static_cast<hipStream_t>(stream), X_dev, N, count, k1, k2, nn, usediters_dev, dir_dev, mu, src_dev);
}

// the device that owns a device pointer, selected (the caller's current device is restored by the DeviceGuard of the entry point)
static nmfx_status select_owner(const void *ptr, const char *what) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) { (void)hipGetLastError(); set_error("%s: not a device pointer", what); return NMFX_ERR_INVALID; }
    return check_device(attr.device);
}
nmfx_status nmfx_minmax_dev(void *stream, const float *X_dev, int64_t count, double *out_dev) {
    if (!X_dev || !out_dev || count <= 0) { set_error("nmfx_minmax_dev: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;   // launch (and allocate the partials) on the device X lives on, whatever the caller's current device is
    TRY(select_owner(X_dev, "nmfx_minmax_dev"));
    return minmax_dev(static_cast<hipStream_t>(stream), X_dev, (long)count, out_dev);
}
nmfx_status nmfx_scale_dev(void *stream, const float *X_dev, int64_t count, double divide_by, float *out_dev) {
    if (!X_dev || !out_dev || count <= 0) { set_error("nmfx_scale_dev: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(select_owner(X_dev, "nmfx_scale_dev"));
    return scale_div(static_cast<hipStream_t>(stream), X_dev, (long)count, divide_by, out_dev);
}

nmfx_status nmfx_projfunc(int64_t N, int32_t count, int32_t dtype, const void *s, double k1, double k2, int32_t nn, void *v,
                          int32_t *usediters, int32_t device) {
    if (N <= 0 || count <= 0 || !s || !v) { set_error("nmfx_projfunc: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(device));
    if (dtype != NMFX_F32 && dtype != NMFX_F64) { set_error("nmfx_projfunc: dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    const size_t tot = (size_t)N * count;
    DevBuf X, it;
    TRY(X.alloc(tot * dsize(dtype))); TRY(it.alloc(sizeof(int) * count));
    hipStream_t st = nullptr;
    StreamDrain drain_(st);
    // the vectors stay in the caller's precision: float64 input is projected in float64 end to end (projfunc.m computes in double)
    NMFX_HIP(hipMemcpyAsync(X.p, s, tot * dsize(dtype), hipMemcpyHostToDevice, st));
    if (dtype == NMFX_F64) TRY(projfunc_cols_f64(st, X.as<double>(), N, count, k1, k2, nn, it.as<int>()));
    else TRY(projfunc_cols(st, X.as<float>(), N, count, k1, k2, nn, it.as<int>()));
    if (usediters) NMFX_HIP(hipMemcpyAsync(usediters, it.p, sizeof(int) * count, hipMemcpyDeviceToHost, st));
    NMFX_HIP(hipMemcpyAsync(v, X.p, tot * dsize(dtype), hipMemcpyDeviceToHost, st));
    NMFX_HIP(hipStreamSynchronize(st));
    return NMFX_OK;
}

}  // extern "C"
