// K10: large-offset pre-aligner = FFT phase correlation
// (skimage.registration.phase_cross_correlation with 0.24 defaults, reference
// matcher/large_offset.py:39; algorithm SURVEY App. B):
//   F = fftn(reference_image), G = fftn(moving_image)  (complex128, as the reference)
//   P = F * conj(G);  P /= max(|P|, 100*eps);  cc = ifftn(P);  (r,c) = first argmax |cc|
//   shift = (r,c), minus N where > fix(N/2).
// Two hand-written transforms and no FFT library: float32 (k_fft.hip) answers when its peak stands clear, the double-precision
// transform (k_fft64.hip) otherwise, under `phase_fp64`, and for the side lengths the float32 kernels do not factor.
#include "common.hpp"

void kp_destroy(km_ctx *c)
{
    c->f64_h = c->f64_w = 0;      // (the tables live in workspace slots, which go with the context)
}

int kp_phase_shift(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                   double out_rc[2])
{
    if (H < 1 || W < 1) return km_fail(c, KM_E_ARG, "phase_shift: empty image");
    c->phase_path = 2; c->phase_margin = 0.0;
    if (!c->opt_phase_fp64 && kp_fast_supported(H, W)) {
        // float32, hand-written FFT (k_fft.hip).  An integer arg-max needs no more precision than that as long as the peak
        // stands clear: when the two largest samples are within 1 % of each other (a shift of exactly x.5 pixels splits the
        // peak evenly, a flat image has none) the double-precision evaluation below decides, as in the reference.
        double margin = 0.0;
        const int rc = kp_phase_shift_fast(c, d_a, d_b, dtype, H, W, stride_a, stride_b, out_rc, &margin);
        c->phase_margin = margin;
        if (rc == KM_OK && margin >= 0.01) { c->phase_path = 1; return KM_OK; }
        if (rc != KM_OK && rc != KM_E_UNSUPPORTED) return rc;
    }
    return kp_phase_shift_f64(c, d_a, d_b, dtype, H, W, stride_a, stride_b, out_rc);
}
