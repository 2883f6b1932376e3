// Host <-> device staging of the blocking (MATLAB-facing) entry points: fp32 / fp64 host buffers in, fp32 on the device.
#include "api_common.h"

namespace nmfx {

// host (f32/f64) -> device fp32, converted on the device through a staging buffer; out = in / divide_by
nmfx_status upload(hipStream_t st, const void *host, int dtype, float *dev, size_t count, double divide_by, DevBuf &stage, size_t stage_elems) {
    const char *h = static_cast<const char *>(host);
    for (size_t off = 0; off < count; off += stage_elems) {
        size_t c = count - off < stage_elems ? count - off : stage_elems;
        NMFX_HIP(hipMemcpyAsync(stage.p, h + off * dsize(dtype), c * dsize(dtype), hipMemcpyHostToDevice, st));
        TRY(cvt_to_f32(st, stage.p, dtype, dev + off, (long)c, divide_by));
        NMFX_HIP(hipStreamSynchronize(st));
    }
    return NMFX_OK;
}
nmfx_status download(hipStream_t st, const float *dev, int dtype, void *host, size_t count, DevBuf &stage, size_t stage_elems) {
    char *h = static_cast<char *>(host);
    if (dtype == NMFX_F32) {
        NMFX_HIP(hipMemcpyAsync(h, dev, count * 4, hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));
        return NMFX_OK;
    }
    for (size_t off = 0; off < count; off += stage_elems) {
        size_t c = count - off < stage_elems ? count - off : stage_elems;
        TRY(cvt_to_f64(st, dev + off, stage.as<double>(), (long)c));
        NMFX_HIP(hipMemcpyAsync(h + off * 8, stage.p, c * 8, hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));
    }
    return NMFX_OK;
}

nmfx_status validate_problem(const nmfx_problem *p, const nmfx_result *r, bool nmfsc, bool need_H_init) {
    if (!p || !r) { set_error("null problem/result"); return NMFX_ERR_INVALID; }
    if (p->m <= 0 || p->n <= 0 || p->K_total <= 0 || p->T <= 0) { set_error("m, n, K_total, T must be positive"); return NMFX_ERR_INVALID; }
    if (!p->V || !p->W_init || (need_H_init && !p->H_init) || !r->W || !r->H || !r->cost) { set_error("V, W_init, H_init, result.W, result.H, result.cost are required"); return NMFX_ERR_INVALID; }
    if (p->dtype != NMFX_F32 && p->dtype != NMFX_F64) { set_error("dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    if (p->maxiter <= 0) { set_error("maxiter must be positive (the wrapper applies the reference default)"); return NMFX_ERR_INVALID; }
    if (!nmfsc) {
        if (p->num_sources < 1) { set_error("num_sources must be >= 1"); return NMFX_ERR_INVALID; }
        if (p->num_sources > 1 && !p->K_s) { set_error("K_s is required when num_sources > 1"); return NMFX_ERR_INVALID; }
        if (p->K_s) {
            long sum = 0;
            for (int s = 0; s < p->num_sources; ++s) { if (p->K_s[s] <= 0) { set_error("K_s entries must be positive"); return NMFX_ERR_INVALID; } sum += p->K_s[s]; }
            if (sum != p->K_total) { set_error("sum(K_s) = %ld != K_total = %d", sum, p->K_total); return NMFX_ERR_INVALID; }
        }
        if (p->divergence == NMFX_DIV_AB && p->alpha == 0 && p->beta == 0) {   // nmf.m:120-122
            set_error("alpha = 0 and beta = 0 is not supported at this time.");
            return NMFX_ERR_INVALID;
        }
    }
    return NMFX_OK;
}

}  // namespace nmfx
