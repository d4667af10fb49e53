/* bgo_impl.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's coupling-flow hot path, one function per reference
 * op-chain, in the reference's own operation order.  Included twice by bgo_oracle.c:
 *   REAL=float  + bgk_detmath.h primitives  -> suffix _f32 (bit-comparable with the HIP kernels)
 *   REAL=double + libm                       -> suffix _f64 (independent high-precision check)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this.
 *
 * Reference citations are relative to /root/reference/bgflow/.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* ------------------------------------------------------------------------------------------
 * Rational-quadratic spline transformer.
 *   parameter unpacking : nn/flow/transformer/spline.py:109-126 (_compute_params)
 *   arithmetic          : nflows.transforms.splines.rational_quadratic_spline (third party, not
 *                         vendored), evaluate/search core corroborated by
 *                         the in-tree copy nn/flow/spline.py:121-188
 *   direction flip      : spline.py:133-144 (bgflow forward = nflows inverse=True),
 *                         spline.py:164-175 (bgflow inverse = nflows inverse=False)
 *   out-of-domain       : spline.py:145-155 -> clamp to [left,right] (flag returned)
 *   reduction           : spline.py:157,188  dlogp.sum(-1, keepdim=True)
 *
 * params row layout (P = 3*K*d + n_nc):  [ w: d*K | h: d*K | s: d*K | s_nc: n_nc ]
 * nc_slot[j] = index into the s_nc block for a non-circular dim j, or -1 for a circular dim
 * (evident intent of spline.py:190-204; identical to the reference for all-/none-circular).
 * bgflow_inverse = 0 : reference _forward (root solve); 1 : reference _inverse (evaluate).
 * Optional outputs (NULL to skip): dlogp_elem [B,d], bin_idx [B,d], knots [B,d,K+1] (search knots,
 * after the in-place +1e-6 on the last one).
 * Returns the number of inputs that were outside [left,right] (clamped).
 * ------------------------------------------------------------------------------------------ */
#define BGO_MAX_BINS 64

int64_t FN(bgo_rqs)(const REAL* y, int64_t ldy, const REAL* params, int64_t ldp,
                    const int32_t* nc_slot, int64_t B, int d, int K, int bgflow_inverse,
                    double left_d, double right_d, double bottom_d, double top_d,
                    double min_w_d, double min_h_d, double min_d_d, int identity_init,
                    REAL* out, int64_t ldo, REAL* dlogp, REAL* dlogp_elem, int32_t* bin_idx,
                    REAL* knots_out)
{
    int64_t n_oob = 0;
    /* beta = ln2 / (1 - min_derivative) is a python double, applied as a scalar of the tensor dtype */
    /* python-double scalars are applied to the tensors in the tensor dtype (torch scalar rule) */
    const REAL beta = (REAL)(identity_init ? (0.6931471805599453 / (1.0 - min_d_d)) : 1.0);
    const REAL w_scale = (REAL)(1.0 - min_w_d * K);
    const REAL h_scale = (REAL)(1.0 - min_h_d * K);
    const REAL min_w = (REAL)min_w_d, min_h = (REAL)min_h_d, min_d = (REAL)min_d_d;
    const REAL left = (REAL)left_d, right = (REAL)right_d, bottom = (REAL)bottom_d, top = (REAL)top_d;
    const REAL xspan = (REAL)(right_d - left_d), yspan = (REAL)(top_d - bottom_d);
#pragma omp parallel for schedule(static) reduction(+ : n_oob)
    for (int64_t b = 0; b < B; ++b) {
        const REAL* prow = params + b * ldp;
        REAL acc = (REAL)0;
        for (int j = 0; j < d; ++j) {
            const REAL* uw = prow + (int64_t)j * K;
            const REAL* uh = prow + (int64_t)d * K + (int64_t)j * K;
            const REAL* us = prow + (int64_t)2 * d * K + (int64_t)j * K;
            REAL cw[BGO_MAX_BINS + 1], ch[BGO_MAX_BINS + 1], W[BGO_MAX_BINS], H[BGO_MAX_BINS];
            /* step 2: widths = softmax; min + (1-min*K)*w; cumsum; pad; affine; ends; diff */
            {
                REAL m = uw[0];
                for (int k = 1; k < K; ++k) m = uw[k] > m ? uw[k] : m;
                REAL e[BGO_MAX_BINS], s = (REAL)0;
                for (int k = 0; k < K; ++k) { e[k] = R_EXP(uw[k] - m); s += e[k]; }
                REAL c = (REAL)0;
                cw[0] = (REAL)0;
                for (int k = 0; k < K; ++k) {
                    REAL wk = e[k] / s;
                    wk = min_w + w_scale * wk;
                    c += wk;
                    cw[k + 1] = c;
                }
                for (int k = 0; k <= K; ++k) cw[k] = xspan * cw[k] + left;
                cw[0] = left; cw[K] = right;
                for (int k = 0; k < K; ++k) W[k] = cw[k + 1] - cw[k];
            }
            /* step 4: same for heights */
            {
                REAL m = uh[0];
                for (int k = 1; k < K; ++k) m = uh[k] > m ? uh[k] : m;
                REAL e[BGO_MAX_BINS], s = (REAL)0;
                for (int k = 0; k < K; ++k) { e[k] = R_EXP(uh[k] - m); s += e[k]; }
                REAL c = (REAL)0;
                ch[0] = (REAL)0;
                for (int k = 0; k < K; ++k) {
                    REAL hk = e[k] / s;
                    hk = min_h + h_scale * hk;
                    c += hk;
                    ch[k + 1] = c;
                }
                for (int k = 0; k <= K; ++k) ch[k] = yspan * ch[k] + bottom;
                ch[0] = bottom; ch[K] = top;
                for (int k = 0; k < K; ++k) H[k] = ch[k + 1] - ch[k];
            }
            /* input, clamped on InputOutsideDomain (spline.py:145-155) */
            REAL x = y[b * ldy + j];
            if (x < left || x > right) { n_oob += 1; x = x < left ? left : (x > right ? right : x); }
            /* step 5: search on cumheights (nflows inverse=True) or cumwidths, in-place +eps */
            REAL* knots = bgflow_inverse ? cw : ch;
            knots[K] += (REAL)1e-6;
            int idx = -1;
            for (int k = 0; k <= K; ++k) idx += (x >= knots[k]) ? 1 : 0;
            if (knots_out) for (int k = 0; k <= K; ++k) knots_out[(b * d + j) * (K + 1) + k] = knots[k];
            if (idx < 0) idx = 0;
            if (idx > K - 1) idx = K - 1;
            if (bin_idx) bin_idx[b * d + j] = idx;
            /* step 3 (only the two gathered derivatives are needed): min_d + softplus(s, beta).
             * slopes[..., K] = slopes[..., 0] (periodic) unless the dim is non-circular. */
            REAL s_lo = us[idx];
            REAL s_hi;
            if (idx + 1 < K) s_hi = us[idx + 1];
            else s_hi = (nc_slot[j] >= 0) ? prow[(int64_t)3 * d * K + nc_slot[j]] : us[0];
            REAL d_i = min_d + R_SOFTPLUS(s_lo, beta);
            REAL d_ip1 = min_d + R_SOFTPLUS(s_hi, beta);
            /* step 6 gathers */
            REAL cw_i = cw[idx], W_i = W[idx], ch_i = ch[idx], H_i = H[idx];
            REAL delta = H_i / W_i;
            REAL o, lad;
            if (!bgflow_inverse) {
                /* step 7: nflows inverse=True */
                REAL dx = x - ch_i;
                REAL S = d_i + d_ip1 - (REAL)2 * delta;
                REAL a = dx * S + H_i * (delta - d_i);
                REAL bb = H_i * d_i - dx * S;
                REAL c = -delta * dx;
                REAL disc = bb * bb - (REAL)4 * a * c;
                REAL root = ((REAL)2 * c) / (-bb - R_SQRT(disc));
                o = root * W_i + cw_i;
                REAL t1mt = root * ((REAL)1 - root);
                REAL den = delta + S * t1mt;
                REAL omr = (REAL)1 - root;
                REAL num = (delta * delta) * (d_ip1 * (root * root) + (REAL)2 * delta * t1mt + d_i * (omr * omr));
                lad = -(R_LOG(num) - (REAL)2 * R_LOG(den));
            } else {
                /* step 8: nflows inverse=False */
                REAL theta = (x - cw_i) / W_i;
                REAL t1mt = theta * ((REAL)1 - theta);
                REAL S = d_i + d_ip1 - (REAL)2 * delta;
                REAL numer = H_i * (delta * (theta * theta) + d_i * t1mt);
                REAL den = delta + S * t1mt;
                o = ch_i + numer / den;
                REAL omt = (REAL)1 - theta;
                REAL num = (delta * delta) * (d_ip1 * (theta * theta) + (REAL)2 * delta * t1mt + d_i * (omt * omt));
                lad = R_LOG(num) - (REAL)2 * R_LOG(den);
            }
            out[b * ldo + j] = o;
            if (dlogp_elem) dlogp_elem[b * d + j] = lad;
            acc += lad;
        }
        dlogp[b] = acc;
    }
    return n_oob;
}

/* ------------------------------------------------------------------------------------------
 * Backward (vector-Jacobian product) of bgo_rqs for first-order losses.  The reference has no
 * backward code: it differentiates the op chain above with torch autograd; this is the analytic
 * VJP of the same chain (softmax -> min + scale*p -> cumsum knots -> gather -> rational-quadratic
 * evaluate / implicit root), pinned against autograd gradients of the reference in
 * tests/test_oracle_golden.py.  The bin index is piecewise constant (no gradient); a clamped
 * (out-of-domain) input gets zero input-gradient like torch.clamp.
 *   g_out [B,d] (ldgo), g_dlogp [B]  ->  g_y [B,d] (ldgy), g_params [B,P] (ldgp, fully written)
 * ------------------------------------------------------------------------------------------ */
void FN(bgo_rqs_backward)(const REAL* y, int64_t ldy, const REAL* params, int64_t ldp,
                          const int32_t* nc_slot, int64_t B, int d, int K, int bgflow_inverse,
                          double left_d, double right_d, double bottom_d, double top_d,
                          double min_w_d, double min_h_d, double min_d_d, int identity_init,
                          const REAL* g_out, int64_t ldgo, const REAL* g_dlogp,
                          REAL* g_y, int64_t ldgy, REAL* g_params, int64_t ldgp)
{
    const REAL beta = (REAL)(identity_init ? (0.6931471805599453 / (1.0 - min_d_d)) : 1.0);
    const REAL w_scale = (REAL)(1.0 - min_w_d * K);
    const REAL h_scale = (REAL)(1.0 - min_h_d * K);
    const REAL min_w = (REAL)min_w_d, min_h = (REAL)min_h_d, min_d = (REAL)min_d_d;
    const REAL left = (REAL)left_d, right = (REAL)right_d, bottom = (REAL)bottom_d, top = (REAL)top_d;
    const REAL xspan = (REAL)(right_d - left_d), yspan = (REAL)(top_d - bottom_d);
    int n_nc = 0;
    for (int j = 0; j < d; ++j) if (nc_slot[j] >= 0) n_nc++;
    const int P = 3 * d * K + n_nc;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const REAL* prow = params + b * ldp;
        REAL* grow = g_params + b * ldgp;
        for (int q = 0; q < P; ++q) grow[q] = (REAL)0;
        for (int j = 0; j < d; ++j) {
            const REAL* uw = prow + (int64_t)j * K;
            const REAL* uh = prow + (int64_t)d * K + (int64_t)j * K;
            const REAL* us = prow + (int64_t)2 * d * K + (int64_t)j * K;
            REAL pw[BGO_MAX_BINS], ph[BGO_MAX_BINS], cw[BGO_MAX_BINS + 1], ch[BGO_MAX_BINS + 1];
            {
                REAL m = uw[0]; for (int k = 1; k < K; ++k) m = uw[k] > m ? uw[k] : m;
                REAL s = (REAL)0; for (int k = 0; k < K; ++k) { pw[k] = R_EXP(uw[k] - m); s += pw[k]; }
                REAL c = (REAL)0; cw[0] = (REAL)0;
                for (int k = 0; k < K; ++k) { pw[k] = pw[k] / s; c += min_w + w_scale * pw[k]; cw[k + 1] = c; }
                for (int k = 0; k <= K; ++k) cw[k] = xspan * cw[k] + left;
                cw[0] = left; cw[K] = right;
            }
            {
                REAL m = uh[0]; for (int k = 1; k < K; ++k) m = uh[k] > m ? uh[k] : m;
                REAL s = (REAL)0; for (int k = 0; k < K; ++k) { ph[k] = R_EXP(uh[k] - m); s += ph[k]; }
                REAL c = (REAL)0; ch[0] = (REAL)0;
                for (int k = 0; k < K; ++k) { ph[k] = ph[k] / s; c += min_h + h_scale * ph[k]; ch[k + 1] = c; }
                for (int k = 0; k <= K; ++k) ch[k] = yspan * ch[k] + bottom;
                ch[0] = bottom; ch[K] = top;
            }
            REAL x = y[b * ldy + j];
            int clamped = 0;
            if (x < left || x > right) { clamped = 1; x = x < left ? left : (x > right ? right : x); }
            const REAL* knots = bgflow_inverse ? cw : ch;
            int idx = -1;
            for (int k = 0; k <= K; ++k) idx += (x >= (k == K ? knots[k] + (REAL)1e-6 : knots[k])) ? 1 : 0;
            if (idx < 0) idx = 0;
            if (idx > K - 1) idx = K - 1;
            const int hi_is_last = (idx + 1 == K);
            const REAL s_lo = us[idx];
            const REAL s_hi = !hi_is_last ? us[idx + 1] : ((nc_slot[j] >= 0) ? prow[(int64_t)3 * d * K + nc_slot[j]] : us[0]);
            const REAL d0 = min_d + R_SOFTPLUS(s_lo, beta), d1 = min_d + R_SOFTPLUS(s_hi, beta);
            const REAL cw_i = cw[idx], W_i = cw[idx + 1] - cw[idx], ch_i = ch[idx], H_i = ch[idx + 1] - ch[idx];
            const REAL delta = H_i / W_i, S = d0 + d1 - (REAL)2 * delta;
            REAL theta;
            if (!bgflow_inverse) {
                REAL dx = x - ch_i;
                REAL a = dx * S + H_i * (delta - d0), bb = H_i * d0 - dx * S, c = -delta * dx;
                theta = ((REAL)2 * c) / (-bb - R_SQRT(bb * bb - (REAL)4 * a * c));
            } else {
                theta = (x - cw_i) / W_i;
            }
            const REAL t = theta * ((REAL)1 - theta), tp = (REAL)1 - (REAL)2 * theta, omt = (REAL)1 - theta;
            const REAL N = delta * theta * theta + d0 * t, den = delta + S * t;
            const REAL Q = N / den;
            const REAL N_th = (REAL)2 * delta * theta + d0 * tp, den_th = S * tp;
            const REAL Q_th = (N_th * den - N * den_th) / (den * den);
            const REAL Q_de = (theta * theta * den - N * ((REAL)1 - (REAL)2 * t)) / (den * den);
            const REAL Q_d0 = (t * den - N * t) / (den * den);
            const REAL Q_d1 = (-N * t) / (den * den);
            const REAL M = d1 * theta * theta + (REAL)2 * delta * t + d0 * omt * omt;
            const REAL lf_th = ((REAL)2 * d1 * theta + (REAL)2 * delta * tp - (REAL)2 * d0 * omt) / M - (REAL)2 * den_th / den;
            const REAL lf_de = (REAL)2 / delta + (REAL)2 * t / M - (REAL)2 * ((REAL)1 - (REAL)2 * t) / den;
            const REAL lf_d0 = omt * omt / M - (REAL)2 * t / den;
            const REAL lf_d1 = theta * theta / M - (REAL)2 * t / den;
            const REAL gy = g_out[b * ldgo + j], gl = g_dlogp[b];
            REAL G_de, G_d0, G_d1, G_H, G_W, G_cw, G_ch, gx;
            if (bgflow_inverse) {
                const REAL G_th = gy * H_i * Q_th + gl * lf_th;
                G_de = gy * H_i * Q_de + gl * lf_de;
                G_d0 = gy * H_i * Q_d0 + gl * lf_d0;
                G_d1 = gy * H_i * Q_d1 + gl * lf_d1;
                G_H = gy * Q + G_de / W_i;
                G_W = -G_de * delta / W_i - G_th * theta / W_i;
                G_ch = gy;
                G_cw = -G_th / W_i;
                gx = G_th / W_i;
            } else {
                const REAL A_th = gy * W_i - gl * lf_th;     /* out = cw_i + W_i theta, lad = -lf */
                const REAL inv = (REAL)1 / (H_i * Q_th);
                G_de = -gl * lf_de - A_th * Q_de / Q_th;
                G_d0 = -gl * lf_d0 - A_th * Q_d0 / Q_th;
                G_d1 = -gl * lf_d1 - A_th * Q_d1 / Q_th;
                G_H = G_de / W_i - A_th * Q * inv;
                G_W = -G_de * delta / W_i + gy * theta;
                G_cw = gy;
                G_ch = -A_th * inv;
                gx = A_th * inv;
            }
            g_y[b * ldgy + j] = clamped ? (REAL)0 : gx;
            /* knots: W_i = cw[i+1] - cw[i] (cw[0], cw[K] are constants) -> W'_m via the cumulative sum -> softmax */
            {
                const REAL gA = (idx >= 1) ? (G_cw - G_W) : (REAL)0, gB = (idx + 1 <= K - 1) ? G_W : (REAL)0;
                REAL gp[BGO_MAX_BINS], dot = (REAL)0;
                for (int m = 0; m < K; ++m) { gp[m] = w_scale * xspan * ((m < idx ? gA : (REAL)0) + (m <= idx ? gB : (REAL)0)); dot += pw[m] * gp[m]; }
                for (int m = 0; m < K; ++m) grow[(int64_t)j * K + m] = pw[m] * (gp[m] - dot);
            }
            {
                const REAL gA = (idx >= 1) ? (G_ch - G_H) : (REAL)0, gB = (idx + 1 <= K - 1) ? G_H : (REAL)0;
                REAL gp[BGO_MAX_BINS], dot = (REAL)0;
                for (int m = 0; m < K; ++m) { gp[m] = h_scale * yspan * ((m < idx ? gA : (REAL)0) + (m <= idx ? gB : (REAL)0)); dot += ph[m] * gp[m]; }
                for (int m = 0; m < K; ++m) grow[(int64_t)d * K + (int64_t)j * K + m] = ph[m] * (gp[m] - dot);
            }
            /* slopes: dD/ds = sigmoid(beta s) (1 beyond the softplus threshold) */
            {
                const REAL z0 = s_lo * beta, z1 = s_hi * beta;
                const REAL sg0 = z0 > (REAL)20 ? (REAL)1 : (REAL)1 / ((REAL)1 + R_EXP(-z0));
                const REAL sg1 = z1 > (REAL)20 ? (REAL)1 : (REAL)1 / ((REAL)1 + R_EXP(-z1));
                grow[(int64_t)2 * d * K + (int64_t)j * K + idx] += G_d0 * sg0;
                if (!hi_is_last) grow[(int64_t)2 * d * K + (int64_t)j * K + idx + 1] += G_d1 * sg1;
                else if (nc_slot[j] >= 0) grow[(int64_t)3 * d * K + nc_slot[j]] += G_d1 * sg1;
                else grow[(int64_t)2 * d * K + (int64_t)j * K] += G_d1 * sg1;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Affine (RealNVP / NICE) transformer: nn/flow/transformer/affine.py:35-70.
 *   mu [B,d] (NULL -> 0), s_raw [B,d] = scale-net output before tanh (NULL -> log_sigma = 0),
 *   log_sigma = tanh(s_raw) * exp(log_alpha)  [- mean over d if preserve_volume]
 *   fwd: y' = exp(log_sigma)*y + mu, dlogp = sum log_sigma ; inv: y' = exp(-log_sigma)*(y-mu),
 *   dlogp = sum(-log_sigma) ; circular: y' = y' mod 1 (python-style, result in [0,1)).
 * ------------------------------------------------------------------------------------------ */
void FN(bgo_affine)(const REAL* y, int64_t ldy, const REAL* mu, int64_t ldmu,
                    const REAL* s_raw, int64_t lds, REAL log_alpha, int preserve_volume,
                    int is_circular, int inverse, int64_t B, int d,
                    REAL* out, int64_t ldo, REAL* dlogp)
{
    const REAL alpha = R_EXP(log_alpha);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        REAL mean = (REAL)0;
        if (s_raw && preserve_volume) {
            REAL s = (REAL)0;
            for (int j = 0; j < d; ++j) s += R_TANH(s_raw[b * lds + j]) * alpha;
            mean = s / (REAL)d;
        }
        REAL acc = (REAL)0;
        for (int j = 0; j < d; ++j) {
            REAL ls = (REAL)0;
            if (s_raw) { ls = R_TANH(s_raw[b * lds + j]) * alpha; if (preserve_volume) ls = ls - mean; }
            REAL m = mu ? mu[b * ldmu + j] : (REAL)0;
            REAL v = y[b * ldy + j];
            REAL o;
            if (!inverse) { o = R_EXP(ls) * v + m; acc += ls; }
            else { o = R_EXP(-ls) * (v - m); acc += -ls; }
            if (is_circular) { o = o - R_TRUNC(o); if (o < (REAL)0) o = o + (REAL)1; } /* torch.remainder(o, 1.0) */
            out[b * ldo + j] = o;
        }
        dlogp[b] = acc;
    }
}

/* Backward (VJP) of bgo_affine; the reference uses torch autograd through affine.py:35-70.
 * g_log_alpha is RETURNED as the sum over the batch (double accumulator). */
double FN(bgo_affine_backward)(const REAL* y, int64_t ldy, const REAL* mu, int64_t ldmu,
                               const REAL* s_raw, int64_t lds, REAL log_alpha, int preserve_volume,
                               int is_circular, int inverse, int64_t B, int d,
                               const REAL* g_out, int64_t ldgo, const REAL* g_dlogp,
                               REAL* g_y, REAL* g_mu, REAL* g_s)
{
    const REAL alpha = R_EXP(log_alpha);
    double g_alpha = 0.0;
    (void)is_circular;   /* d(o mod 1)/do = 1 */
#pragma omp parallel for schedule(static) reduction(+ : g_alpha)
    for (int64_t b = 0; b < B; ++b) {
        REAL mean = (REAL)0;
        if (s_raw && preserve_volume) {
            for (int j = 0; j < d; ++j) mean += R_TANH(s_raw[b * lds + j]) * alpha;
            mean = mean / (REAL)d;
        }
        REAL gmean = (REAL)0;
        for (int pass = 0; pass < 2; ++pass) {
            /* pass 0: accumulate mean of g_ls (only needed for preserve_volume); pass 1: write */
            if (pass == 0 && !(s_raw && preserve_volume)) continue;
            REAL acc = (REAL)0;
            for (int j = 0; j < d; ++j) {
                REAL th = s_raw ? R_TANH(s_raw[b * lds + j]) : (REAL)0;
                REAL ls = s_raw ? th * alpha - mean : (REAL)0;
                REAL m = mu ? mu[b * ldmu + j] : (REAL)0;
                REAL v = y[b * ldy + j], go = g_out[b * ldgo + j], gl = g_dlogp[b];
                REAL gy, gm, gls;
                if (!inverse) { REAL e = R_EXP(ls); gy = go * e; gm = go; gls = go * e * v + gl; }
                else { REAL e = R_EXP(-ls); gy = go * e; gm = -go * e; gls = -go * e * (v - m) - gl; }
                if (pass == 0) { acc += gls; continue; }
                REAL gl_raw = gls - gmean;
                g_y[b * d + j] = gy;
                if (g_mu) g_mu[b * d + j] = gm;
                if (g_s) g_s[b * d + j] = gl_raw * alpha * ((REAL)1 - th * th);
                g_alpha += (double)(gl_raw * th);
            }
            if (pass == 0) gmean = acc / (REAL)d;
        }
    }
    return s_raw ? g_alpha * (double)alpha : 0.0;
}

/* ------------------------------------------------------------------------------------------
 * DenseNet conditioner (nn/dense.py:30-48): y = act(... act(x W0^T + b0) ...) W_last^T + b_last.
 * Each output is a k-ascending fma chain starting from 0, bias added last -- the order the
 * gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32) accumulates in, so the f32 flavour is bit-comparable
 * with the fused HIP coupling kernel.  W is torch Linear layout [n_out, n_in] row-major.
 * act: 0 none, 1 SiLU, 2 ReLU, 3 Tanh.   Optional WrapPeriodic featuriser (nn/periodic.py:30-37)
 * is applied by the caller (bgo_wrap_periodic).
 * ------------------------------------------------------------------------------------------ */
void FN(bgo_linear)(const REAL* x, int64_t ldx, const REAL* W, const REAL* bias,
                    int64_t B, int n_in, int n_out, int act, const int32_t* k_order,
                    REAL* out, int64_t ldo)
{
    /* k_order (NULL = 0,1,2,...): the order in which the inputs enter each output's fma chain.  The
     * fused HIP kernel keeps hidden activations in MFMA accumulator registers and feeds them back
     * as the B operand without a shuffle, which visits the 32 features of a block in the order
     * 0,4,1,5,2,6,3,7, 8,12,9,13,...; passing that permutation makes this function bit-identical. */
    /* W^T [n_in][n_out] so that the inner loop runs over outputs: every output is still its own
     * k-ascending fma chain (same bits as the textbook loop), but the loop vectorises. */
    REAL* Wt = (REAL*)malloc(sizeof(REAL) * (size_t)n_in * (size_t)n_out);
    for (int o = 0; o < n_out; ++o)
        for (int k = 0; k < n_in; ++k) Wt[(size_t)k * n_out + o] = W[(size_t)o * n_in + k];
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const REAL* xr = x + b * ldx;
        REAL* acc = out + b * ldo;
        for (int o = 0; o < n_out; ++o) acc[o] = (REAL)0;
        for (int kk = 0; kk < n_in; ++kk) {
            const int k = k_order ? k_order[kk] : kk;
            const REAL xk = xr[k];
            const REAL* wr = Wt + (size_t)k * n_out;
#pragma omp simd
            for (int o = 0; o < n_out; ++o) acc[o] = R_FMA(xk, wr[o], acc[o]);
        }
        for (int o = 0; o < n_out; ++o) {
            REAL v = acc[o];
            if (bias) v = v + bias[o];
            if (act == 1) v = R_SILU(v);
            else if (act == 2) v = v > (REAL)0 ? v : (REAL)0;
            else if (act == 3) v = R_TANH(v);
            acc[o] = v;
        }
    }
    free(Wt);
}

/* WrapPeriodic featuriser, nn/periodic.py:30-37 with all indices periodic on [0,1]:
 * out = [cos(2 pi x), sin(2 pi x)]  ([B,2d]).  f32 flavour: deterministic bgk_sincos2pif (what the
 * fused HIP kernel evaluates); f64 flavour: libm. */
void FN(bgo_wrap_periodic)(const REAL* x, int64_t ldx, int64_t B, int d, REAL* out, int64_t ldo)
{
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int j = 0; j < d; ++j) {
            REAL sv, cv;
            R_SINCOS2PI(x[b * ldx + j], &sv, &cv);
            out[b * ldo + j] = cv;
            out[b * ldo + d + j] = sv;
        }
}

/* ------------------------------------------------------------------------------------------
 * Internal coordinates.
 * ------------------------------------------------------------------------------------------ */
static inline void FN(v3sub)(const REAL* a, const REAL* b, REAL* o) { o[0]=a[0]-b[0]; o[1]=a[1]-b[1]; o[2]=a[2]-b[2]; }
static inline REAL FN(v3dot)(const REAL* a, const REAL* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline void FN(v3cross)(const REAL* a, const REAL* b, REAL* o) {
    o[0] = a[1]*b[2] - a[2]*b[1]; o[1] = a[2]*b[0] - a[0]*b[2]; o[2] = a[0]*b[1] - a[1]*b[0];
}
static inline REAL FN(v3norm)(const REAL* a) { return R_SQRT(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }
static inline REAL FN(det3)(const REAL* r0, const REAL* r1, const REAL* r2) {
    /* ic_helper.py:109-111  (a0 x a1) . a2 */
    REAL c[3]; FN(v3cross)(r0, r1, c); return FN(v3dot)(c, r2);
}

/* xyz -> (bonds, angles, torsions, x_fixed), RelativeInternalCoordinateTransformation._forward
 * (nn/flow/crd_transform/ic.py:386-433) with dist_deriv / angle_deriv / torsion_deriv
 * (ic_helper.py:148-293).  zmat [n,4] int32 rows (a,b,c,d); fixed [n_fixed] atom ids.
 * If whiten != NULL (MixedCoordinateTransformation._forward, ic.py:838-860 + pca.py:74-83):
 * z_fixed = (x_fixed - mean) @ Twhiten  ([3*n_fixed, keep] row-major), dlogp += jac_xz. */
void FN(bgo_ic_xyz2ic)(const REAL* x, int64_t ldx, const int32_t* zmat, int n, const int32_t* fixed,
                       int n_fixed, int normalize, REAL eps, int enforce,
                       const REAL* wh_mean, const REAL* Twhiten, int keep, REAL jac_xz,
                       int64_t B, REAL* bonds, REAL* angles, REAL* torsions, REAL* xfix_out,
                       REAL* dlogp)
{
    const REAL PI = (REAL)3.14159265358979323846;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const REAL* xr = x + b * ldx;
        REAL acc = (REAL)0;
        for (int i = 0; i < n; ++i) {
            const REAL* x1 = xr + 3 * zmat[4*i+0];
            const REAL* x2 = xr + 3 * zmat[4*i+1];
            const REAL* x3 = xr + 3 * zmat[4*i+2];
            const REAL* x4 = xr + 3 * zmat[4*i+3];
            /* dist_deriv(x1, x2) */
            REAL r[3]; FN(v3sub)(x2, x1, r);
            REAL rn = FN(v3norm)(r); if (enforce && rn < eps) rn = eps;
            REAL Jb[3] = { -r[0]/rn, -r[1]/rn, -r[2]/rn };
            bonds[b * n + i] = rn;
            /* angle_deriv(x1, x2, x3) */
            REAL r12[3]; FN(v3sub)(x1, x2, r12);
            REAL n12 = FN(v3norm)(r12); if (enforce && n12 < eps) n12 = eps;
            REAL rn12[3] = { r12[0]/n12, r12[1]/n12, r12[2]/n12 };
            REAL r32[3]; FN(v3sub)(x3, x2, r32);
            REAL n32 = FN(v3norm)(r32); if (enforce && n32 < eps) n32 = eps;
            REAL rn32[3] = { r32[0]/n32, r32[1]/n32, r32[2]/n32 };
            REAL cosa = FN(v3dot)(rn12, rn32);
            /* J = rn32^T (I - rn12 rn12^T) / n12 */
            REAL Ja[3];
            for (int c = 0; c < 3; ++c) {
                REAL s = (REAL)0;
                for (int k = 0; k < 3; ++k) {
                    REAL Pkc = ((k == c ? (REAL)1 : (REAL)0) - rn12[k] * rn12[c]) / n12;
                    s += rn32[k] * Pkc;
                }
                Ja[c] = s;
            }
            if (enforce) { if (cosa < (REAL)-1 + eps) cosa = (REAL)-1 + eps; if (cosa > (REAL)1 - eps) cosa = (REAL)1 - eps; }
            REAL ang = R_ACOS(cosa);
            REAL sq = R_SQRT((REAL)1 - cosa * cosa);
            for (int c = 0; c < 3; ++c) Ja[c] = -Ja[c] / sq;
            /* torsion_deriv(x1, x2, x3, x4) */
            REAL b0[3] = { -(x2[0]-x1[0]), -(x2[1]-x1[1]), -(x2[2]-x1[2]) };
            REAL b1[3]; FN(v3sub)(x3, x2, b1);
            REAL b2[3]; FN(v3sub)(x4, x3, b2);
            REAL b1n = FN(v3norm)(b1); if (enforce && b1n < eps) b1n = eps;
            REAL u[3] = { b1[0]/b1n, b1[1]/b1n, b1[2]/b1n };
            REAL b0u = FN(v3dot)(b0, u), b2u = FN(v3dot)(b2, u);
            REAL v[3] = { b0[0]-b0u*u[0], b0[1]-b0u*u[1], b0[2]-b0u*u[2] };
            REAL w[3] = { b2[0]-b2u*u[0], b2[1]-b2u*u[1], b2[2]-b2u*u[2] };
            REAL xx = FN(v3dot)(v, w);
            REAL uxv[3]; FN(v3cross)(u, v, uxv);
            REAL yy = FN(v3dot)(uxv, w);
            REAL tor = R_ATAN2(yy, xx);
            REAL q = xx * xx + yy * yy; if (enforce && q < eps) q = eps;
            REAL dadx = -yy / q, dady = xx / q;
            /* J = dadx * w^T P + dady * w^T A P, P = I - u u^T, A = skew(u) with (A v) = u x v.
             * w^T A = (w x u)^T   [row vector times skew matrix of ic_helper.py:86-101] */
            REAL wxu[3]; FN(v3cross)(w, u, wxu);
            REAL g[3] = { dadx * w[0] + dady * wxu[0], dadx * w[1] + dady * wxu[1], dadx * w[2] + dady * wxu[2] };
            REAL gu = FN(v3dot)(g, u);
            REAL Jt[3] = { g[0] - gu * u[0], g[1] - gu * u[1], g[2] - gu * u[2] };
            REAL det = FN(det3)(Jb, Ja, Jt);
            acc += R_LOG(R_FABS(det));
            if (normalize) { ang = ang / PI; tor = (tor + PI) / ((REAL)2 * PI); }
            angles[b * n + i] = ang;
            torsions[b * n + i] = tor;
        }
        if (normalize) acc += -(REAL)n * R_LOG(PI) - (REAL)n * R_LOG((REAL)2 * PI);
        const int nf3 = 3 * n_fixed;
        if (Twhiten) {
            for (int k = 0; k < keep; ++k) {
                REAL s = (REAL)0;
                for (int c = 0; c < nf3; ++c) {
                    REAL xc = xr[3 * fixed[c / 3] + c % 3] - wh_mean[c];
                    s += xc * Twhiten[c * keep + k];
                }
                xfix_out[b * keep + k] = s;
            }
            acc += jac_xz;
        } else {
            for (int c = 0; c < nf3; ++c) xfix_out[b * nf3 + c] = xr[3 * fixed[c / 3] + c % 3];
        }
        dlogp[b] = acc;
    }
}

/* (bonds, angles, torsions, x_fixed) -> xyz, RelativeInternalCoordinateTransformation._inverse
 * (ic.py:435-513) with ic2xyz_deriv (ic_helper.py:372-452).  place [n,5] int32 rows in placement
 * order (decompose_z_matrix, ic.py:25-91): (atom, p1, p2, p3, zrow) -- atom ids index the OUTPUT
 * atom order directly (equivalent to the reference's index2atom/atom2index bookkeeping and the
 * final permutation ic.py:511).  If Tblacken != NULL (Mixed _inverse, ic.py:862-884 +
 * pca.py:85-93): x_fixed = z_fixed @ Tblacken + mean ([keep, 3*n_fixed]), dlogp -= jac_xz. */
void FN(bgo_ic_ic2xyz)(const REAL* bonds, const REAL* angles, const REAL* torsions, const REAL* xfix,
                       const int32_t* place, int n, const int32_t* fixed, int n_fixed,
                       int normalize, REAL eps, int enforce,
                       const REAL* wh_mean, const REAL* Tblacken, int keep, REAL jac_xz,
                       int64_t B, REAL* x, int64_t ldx, REAL* dlogp)
{
    const REAL PI = (REAL)3.14159265358979323846;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        REAL* xr = x + b * ldx;
        REAL acc = (REAL)0;
        const int nf3 = 3 * n_fixed;
        if (Tblacken) {
            for (int c = 0; c < nf3; ++c) {
                REAL s = (REAL)0;
                for (int k = 0; k < keep; ++k) s += xfix[b * keep + k] * Tblacken[k * nf3 + c];
                xr[3 * fixed[c / 3] + c % 3] = s + wh_mean[c];
            }
            acc += -jac_xz;
        } else {
            for (int c = 0; c < nf3; ++c) xr[3 * fixed[c / 3] + c % 3] = xfix[b * nf3 + c];
        }
        if (normalize) acc += (REAL)n * R_LOG(PI) + (REAL)n * R_LOG((REAL)2 * PI);
        for (int i = 0; i < n; ++i) {
            const int32_t* pl = place + 5 * i;
            const REAL* p1 = xr + 3 * pl[1];
            const REAL* p2 = xr + 3 * pl[2];
            const REAL* p3 = xr + 3 * pl[3];
            int zr = pl[4];
            REAL dd = bonds[b * n + zr], a = angles[b * n + zr], t = torsions[b * n + zr];
            if (normalize) { a = a * PI; t = t * ((REAL)2 * PI) - PI; }
            REAL v1[3], v2[3], nv[3], nn[3];
            FN(v3sub)(p1, p2, v1); FN(v3sub)(p1, p3, v2);
            FN(v3cross)(v1, v2, nv); FN(v3cross)(v1, nv, nn);
            REAL nvn = FN(v3norm)(nv); if (enforce && nvn < eps) nvn = eps;
            REAL nnn = FN(v3norm)(nn); if (enforce && nnn < eps) nnn = eps;
            REAL nh[3] = { nv[0]/nvn, nv[1]/nvn, nv[2]/nvn };
            REAL nnh[3] = { nn[0]/nnn, nn[1]/nnn, nn[2]/nnn };
            REAL st = R_SIN(t), ct = R_COS(t), sa = R_SIN(a), ca = R_COS(a);
            REAL v3[3] = { nh[0]*(-st) + nnh[0]*ct, nh[1]*(-st) + nnh[1]*ct, nh[2]*(-st) + nnh[2]*ct };
            REAL v3n = FN(v3norm)(v3); if (enforce && v3n < eps) v3n = eps;
            REAL v3h[3] = { v3[0]/v3n, v3[1]/v3n, v3[2]/v3n };
            REAL v1n = FN(v3norm)(v1); if (enforce && v1n < eps) v1n = eps;
            REAL v1h[3] = { v1[0]/v1n, v1[1]/v1n, v1[2]/v1n };
            REAL pos[3], Jd[3], Ja[3], Jt3[3], Jt[3];
            for (int c = 0; c < 3; ++c) {
                pos[c] = p1[c] + v3h[c] * dd * sa - v1h[c] * dd * ca;
                Jd[c] = v3h[c] * sa - v1h[c] * ca;
                Ja[c] = v3h[c] * dd * ca + v1h[c] * dd * sa;
                Jt3[c] = nh[c] * (-ct) + nnh[c] * (-st);
            }
            REAL jt1 = dd * sa;
            REAL h3 = FN(v3dot)(v3h, Jt3);
            for (int c = 0; c < 3; ++c) Jt[c] = jt1 * ((REAL)1 / v3n) * (Jt3[c] - v3h[c] * h3);
            /* J = stack([Jd, Ja, Jt], dim=-1): rows of J are (Jd[c], Ja[c], Jt[c]) (ic_helper.py:450) */
            REAL R0[3] = { Jd[0], Ja[0], Jt[0] }, R1[3] = { Jd[1], Ja[1], Jt[1] }, R2[3] = { Jd[2], Ja[2], Jt[2] };
            REAL det = FN(det3)(R0, R1, R2);
            acc += R_LOG(R_FABS(det));
            REAL* po = xr + 3 * pl[0];
            po[0] = pos[0]; po[1] = pos[1]; po[2] = pos[2];
        }
        dlogp[b] = acc;
    }
}

/* Global reference frame of the first three atoms (ReferenceSystemTransformation, ic.py:128-265;
 * init_xyz2ics / init_ics2xyz, tripod, _to_euler_angles, _from_euler_angles: ic_helper.py:114-138,
 * 330-368, 480-680).  The reference obtains log|det J_9x9| from a batched autograd Jacobian and a
 * 24-term permutation expansion; analytically it is  -(2 ln d01 + 2 ln d12 + ln sin a012)  for
 * xyz -> ICs (beta is stored as a cosine, which absorbs the sin(beta) of the Euler measure) -- checked
 * against the reference to 1e-15 by the golden tests.
 *   fwd:  x3 [B,9] = (x0,x1,x2)  ->  out [B,9] = (x0[3], d01, d12, a012, alpha, beta, gamma), dlogp
 *   inv:  the reverse.  normalize: a012/pi, (alpha,gamma + pi)/2pi; beta stays in [-1,1]. */
void FN(bgo_refsys)(const REAL* in, int64_t B, int inverse, int normalize, REAL eps, int enforce,
                    REAL* out, REAL* dlogp)
{
    const REAL PI = (REAL)3.14159265358979323846;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const REAL* v = in + 9 * b;
        REAL* o = out + 9 * b;
        if (!inverse) {
            const REAL* x0 = v; const REAL* x1 = v + 3; const REAL* x2 = v + 6;
            REAL r01[3], r12[3]; FN(v3sub)(x1, x0, r01); FN(v3sub)(x2, x1, r12);
            REAL d01 = FN(v3norm)(r01); if (enforce && d01 < eps) d01 = eps;
            REAL d12 = FN(v3norm)(r12); if (enforce && d12 < eps) d12 = eps;
            /* angle_deriv(x0, x1, x2): angle at x1 */
            REAL a[3], c[3]; FN(v3sub)(x0, x1, a); FN(v3sub)(x2, x1, c);
            REAL an = FN(v3norm)(a); if (enforce && an < eps) an = eps;
            REAL cn = FN(v3norm)(c); if (enforce && cn < eps) cn = eps;
            REAL cosang = (a[0]/an)*(c[0]/cn) + (a[1]/an)*(c[1]/cn) + (a[2]/an)*(c[2]/cn);
            if (enforce) { if (cosang < (REAL)-1 + eps) cosang = (REAL)-1 + eps; if (cosang > (REAL)1 - eps) cosang = (REAL)1 - eps; }
            REAL a012 = R_ACOS(cosang);
            /* tripod(x0, x1, x2) */
            REAL e1n = FN(v3norm)(r01); if (enforce && e1n < eps) e1n = eps;
            REAL e1[3] = { r01[0]/e1n, r01[1]/e1n, r01[2]/e1n };
            REAL u[3]; FN(v3sub)(x2, x0, u);
            REAL e2[3]; FN(v3cross)(u, e1, e2);
            REAL e2n = FN(v3norm)(e2); if (enforce && e2n < eps) e2n = eps;
            e2[0] /= e2n; e2[1] /= e2n; e2[2] /= e2n;
            REAL e3[3]; FN(v3cross)(e2, e1, e3);
            /* basis (x, y, z) = (-e3, -e2, e1) */
            REAL alpha = R_ATAN2(e1[0], -e1[1]);
            REAL beta = e1[2];
            REAL gamma = R_ATAN2(-e3[2], -e2[2]);
            REAL dl = -((REAL)2 * R_LOG(d01) + (REAL)2 * R_LOG(d12) + R_LOG(R_SIN(a012)));
            if (normalize) {
                a012 = a012 / PI; alpha = (alpha + PI) / ((REAL)2 * PI); gamma = (gamma + PI) / ((REAL)2 * PI);
                dl += -R_LOG(PI) - (REAL)2 * R_LOG((REAL)2 * PI);
            }
            o[0] = x0[0]; o[1] = x0[1]; o[2] = x0[2]; o[3] = d01; o[4] = d12; o[5] = a012; o[6] = alpha; o[7] = beta; o[8] = gamma;
            dlogp[b] = dl;
        } else {
            const REAL* x0 = v;
            REAL d01 = v[3], d12 = v[4], a012 = v[5], alpha = v[6], beta = v[7], gamma = v[8];
            REAL dl = (REAL)0;
            if (normalize) {
                alpha = alpha * ((REAL)2 * PI) - PI; gamma = gamma * ((REAL)2 * PI) - PI; a012 = a012 * PI;
                dl += R_LOG(PI) + (REAL)2 * R_LOG((REAL)2 * PI);
            }
            dl += (REAL)2 * R_LOG(d01) + (REAL)2 * R_LOG(d12) + R_LOG(R_SIN(a012));
            /* local frame: p0 = 0, p1 = (0,0,d01), p2 from ic2xyz(p1, p0, (0,-1,0), d12, a012, pi/2):
             * v1 = p1 - p0 = (0,0,d01) -> v1h = z;  n = v1 x (p1 - p3) ... evaluates to p2 = p1 + d12 (sin a * y' - cos a * z)
             * with the reference's conventions y' = (0, 1, 0) direction below */
            REAL p1[3] = { 0, 0, d01 };
            REAL p3[3] = { 0, -1, 0 }, p0[3] = { 0, 0, 0 };
            REAL v1[3], v2[3], nv[3], nn[3];
            FN(v3sub)(p1, p0, v1); FN(v3sub)(p1, p3, v2);
            FN(v3cross)(v1, v2, nv); FN(v3cross)(v1, nv, nn);
            REAL nvn = FN(v3norm)(nv); if (enforce && nvn < eps) nvn = eps;
            REAL nnn = FN(v3norm)(nn); if (enforce && nnn < eps) nnn = eps;
            REAL tq = (REAL)0.5 * PI, st = R_SIN(tq), ct = R_COS(tq), sa = R_SIN(a012), ca = R_COS(a012);
            REAL v3[3] = { nv[0]/nvn*(-st) + nn[0]/nnn*ct, nv[1]/nvn*(-st) + nn[1]/nnn*ct, nv[2]/nvn*(-st) + nn[2]/nnn*ct };
            REAL v3n = FN(v3norm)(v3); if (enforce && v3n < eps) v3n = eps;
            REAL v1n = FN(v3norm)(v1); if (enforce && v1n < eps) v1n = eps;
            REAL p2[3];
            for (int c = 0; c < 3; ++c) p2[c] = p1[c] + v3[c]/v3n * d12 * sa - v1[c]/v1n * d12 * ca;
            /* R = Rz(alpha) Rx(acos beta) Rz(gamma) (_rotmat3x3 axis 2, 0, 2) */
            REAL bA = R_ACOS(beta);
            REAL caA = R_COS(alpha), saA = R_SIN(alpha), cb = R_COS(bA), sb = R_SIN(bA), cg = R_COS(gamma), sg = R_SIN(gamma);
            REAL Rz1[9] = { caA, -saA, 0, saA, caA, 0, 0, 0, 1 };
            REAL Rx[9] = { 1, 0, 0, 0, cb, -sb, 0, sb, cb };
            REAL Rz2[9] = { cg, -sg, 0, sg, cg, 0, 0, 0, 1 };
            REAL T[9], R[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { REAL s = 0; for (int k = 0; k < 3; ++k) s += Rz1[3*i+k] * Rx[3*k+j]; T[3*i+j] = s; }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { REAL s = 0; for (int k = 0; k < 3; ++k) s += T[3*i+k] * Rz2[3*k+j]; R[3*i+j] = s; }
            for (int e = 0; e < 3; ++e) {
                REAL s1 = 0, s2 = 0;
                for (int dd = 0; dd < 3; ++dd) { s1 += p1[dd] * R[3*e+dd]; s2 += p2[dd] * R[3*e+dd]; }
                o[e] = x0[e]; o[3 + e] = s1 + x0[e]; o[6 + e] = s2 + x0[e];
            }
            dlogp[b] = dl;
        }
    }
}

/* Backward (VJP) of bgo_ic_ic2xyz for first-order losses.  The reference differentiates
 * ic2xyz_deriv (ic_helper.py:372-452) and det3x3(J).abs().log() (ic.py:503) with torch autograd;
 * this is the hand-derived reverse sweep over the placement table, using the analytic identity
 * log|det J| = 2 ln d + ln|sin a| for the log-det term (exact away from the eps clamps).
 * NOT valid for a placement whose norms the forward clamped (atoms within ~3e-4 nm of each other: |v1 x (v1 x v2)| < eps): there the
 * clamped vectors are no unit vectors and the reference's autograd sees torch.clamp's zero derivative -- this sweep is then off by
 * O(1) in either precision.  The checker for such samples is f64 autograd of oracle/torch_flow.py::ic2xyz_torch (the reference's op
 * chain); the HIP kernels evaluate that case on dual numbers (tests/test_gpu_round5.py::test_ic_backward_where_the_forward_clamped_a_norm).
 *   x [B,3*n_atoms] = the forward OUTPUT (saved), g_x [B,3*n_atoms], g_dlogp [B]
 *   -> g_bonds, g_angles, g_torsions [B,n], g_xfix [B,keep] */
void FN(bgo_ic_ic2xyz_backward)(const REAL* bonds, const REAL* angles, const REAL* torsions,
                                const REAL* x, int64_t ldx, const int32_t* place, int n,
                                const int32_t* fixed, int n_fixed, int normalize,
                                const REAL* Tblacken, int keep, int64_t B,
                                const REAL* g_x, int64_t ldgx, const REAL* g_dlogp,
                                REAL* g_bonds, REAL* g_angles, REAL* g_torsions, REAL* g_xfix)
{
    const REAL PI = (REAL)3.14159265358979323846;
    const int n_atoms = n + n_fixed;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const REAL* xr = x + b * ldx;
        REAL* gp = (REAL*)malloc(sizeof(REAL) * 3 * (size_t)n_atoms);      /* any molecule size */
        for (int c = 0; c < 3 * n_atoms; ++c) gp[c] = g_x[b * ldgx + c];
        const REAL gl = g_dlogp[b];
        for (int i = n - 1; i >= 0; --i) {
            const int32_t* pl = place + 5 * i;
            const REAL* p1 = xr + 3 * pl[1]; const REAL* p2 = xr + 3 * pl[2]; const REAL* p3 = xr + 3 * pl[3];
            const int zr = pl[4];
            REAL dd = bonds[b * n + zr], a = angles[b * n + zr], t = torsions[b * n + zr];
            if (normalize) { a = a * PI; t = t * ((REAL)2 * PI) - PI; }
            const REAL* g = gp + 3 * pl[0];
            REAL v1[3], v2[3], nv[3], nn[3];
            FN(v3sub)(p1, p2, v1); FN(v3sub)(p1, p3, v2);
            FN(v3cross)(v1, v2, nv); FN(v3cross)(v1, nv, nn);
            const REAL nvn = FN(v3norm)(nv), nnn = FN(v3norm)(nn), v1n = FN(v3norm)(v1);
            REAL nh[3], nnh[3], v1h[3], v3[3], v3h[3];
            const REAL st = R_SIN(t), ct = R_COS(t), sa = R_SIN(a), ca = R_COS(a);
            for (int c = 0; c < 3; ++c) { nh[c] = nv[c] / nvn; nnh[c] = nn[c] / nnn; v1h[c] = v1[c] / v1n; }
            for (int c = 0; c < 3; ++c) v3[c] = -st * nh[c] + ct * nnh[c];
            const REAL v3n = FN(v3norm)(v3);
            for (int c = 0; c < 3; ++c) v3h[c] = v3[c] / v3n;
            /* scalars */
            REAL gd = (REAL)0, ga = (REAL)0;
            for (int c = 0; c < 3; ++c) {
                gd += g[c] * (sa * v3h[c] - ca * v1h[c]);
                ga += g[c] * (dd * ca * v3h[c] + dd * sa * v1h[c]);
            }
            gd += gl * (REAL)2 / dd;
            ga += gl * ca / sa;
            /* vectors */
            REAL g_v3h[3], g_v1h[3], g_v3[3], g_nh[3], g_nnh[3], g_n[3], g_nn[3], g_v1[3], g_v2[3], tmp[3];
            for (int c = 0; c < 3; ++c) { g_v3h[c] = dd * sa * g[c]; g_v1h[c] = -dd * ca * g[c]; }
            REAL pr = FN(v3dot)(v3h, g_v3h);
            for (int c = 0; c < 3; ++c) g_v3[c] = (g_v3h[c] - v3h[c] * pr) / v3n;
            REAL gt = (REAL)0;
            for (int c = 0; c < 3; ++c) { gt += g_v3[c] * (-ct * nh[c] - st * nnh[c]); g_nh[c] = -st * g_v3[c]; g_nnh[c] = ct * g_v3[c]; }
            pr = FN(v3dot)(nh, g_nh);
            for (int c = 0; c < 3; ++c) g_n[c] = (g_nh[c] - nh[c] * pr) / nvn;
            pr = FN(v3dot)(nnh, g_nnh);
            for (int c = 0; c < 3; ++c) g_nn[c] = (g_nnh[c] - nnh[c] * pr) / nnn;
            /* nn = v1 x n :  g_v1 = n x g_nn ; g_n += g_nn x v1 */
            FN(v3cross)(nv, g_nn, g_v1);
            FN(v3cross)(g_nn, v1, tmp); for (int c = 0; c < 3; ++c) g_n[c] += tmp[c];
            /* n = v1 x v2 :  g_v1 += v2 x g_n ; g_v2 = g_n x v1 */
            FN(v3cross)(v2, g_n, tmp); for (int c = 0; c < 3; ++c) g_v1[c] += tmp[c];
            FN(v3cross)(g_n, v1, g_v2);
            pr = FN(v3dot)(v1h, g_v1h);
            for (int c = 0; c < 3; ++c) g_v1[c] += (g_v1h[c] - v1h[c] * pr) / v1n;
            for (int c = 0; c < 3; ++c) {
                gp[3 * pl[1] + c] += g[c] + g_v1[c] + g_v2[c];
                gp[3 * pl[2] + c] -= g_v1[c];
                gp[3 * pl[3] + c] -= g_v2[c];
            }
            if (normalize) { ga = ga * PI; gt = gt * ((REAL)2 * PI); }
            g_bonds[b * n + zr] = gd; g_angles[b * n + zr] = ga; g_torsions[b * n + zr] = gt;
        }
        const int nf3 = 3 * n_fixed;
        if (Tblacken) {
            for (int k = 0; k < keep; ++k) {
                REAL s = (REAL)0;
                for (int c = 0; c < nf3; ++c) s += gp[3 * fixed[c / 3] + c % 3] * Tblacken[k * nf3 + c];
                g_xfix[b * keep + k] = s;
            }
        } else {
            for (int c = 0; c < nf3; ++c) g_xfix[b * nf3 + c] = gp[3 * fixed[c / 3] + c % 3];
        }
        free(gp);
    }
}

#undef FN
#undef CAT
#undef CAT_
