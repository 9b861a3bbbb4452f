/*
 * drr_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see drr_oracle.c).
 *
 * Included twice by drr_oracle.c, once with REAL=float and once with
 * REAL=double.  Every function is a per-ray, loop-by-loop restatement of the
 * vectorised tensor program in the reference's diffdrr/renderers.py; the
 * arithmetic is kept in the reference's operation order so that the float
 * build reproduces the reference's fp32 rounding as closely as a scalar
 * program can (the only deliberate difference: the final reduction over a
 * ray's terms is accumulated in double, which is at least as accurate as
 * torch.sum's pairwise fp32 reduction).
 *
 * Conventions shared with the reference:
 *   volume  V[Dx][Dy][Dz], C-contiguous (z fastest)    drr.py:81-85
 *   source  (B, src_n, 3) with src_n == 1 (broadcast) or N
 *   target  (B, N, 3)      voxel-index coordinates      drr.py:204-205
 *   img     (B, N)         ray length in world units    drr.py:201
 *   out     (B, N)
 */

#ifndef REAL
#error "define REAL and SFX before including"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

/* ------------------------------------------------------------------ helpers */

/* grid_sample(align_corners=False) un-normalisation of a coordinate that
 * _get_xyzs normalised: renderers.py:152 then aten grid_sampler
 * `((g + 1) * size - 1) / 2`.  align_corners=True: `(g + 1) / 2 * (size-1)`. */
static inline REAL FN(grid_coord_)(REAL x, REAL shift, int D, int align_corners) {
    REAL g = (REAL)2 * (x + shift) / (REAL)D - (REAL)1; /* renderers.py:152 */
    if (align_corners) return (g + (REAL)1) / (REAL)2 * (REAL)(D - 1);
    return ((g + (REAL)1) * (REAL)D - (REAL)1) / (REAL)2;
}

/* aten grid_sampler_3d "nearest": nearbyint per axis, zeros outside.
 * The reference passes volume.permute(2,1,0) (renderers.py:160) so grid
 * component 0 indexes volume axis 0. */
static inline REAL FN(fetch_nearest_)(const REAL *V, int Dx, int Dy, int Dz, REAL gx, REAL gy,
                                      REAL gz, long *flat) {
    /* rint() rounds half to even in the default rounding mode == nearbyint */
    double rx = rint((double)gx), ry = rint((double)gy), rz = rint((double)gz);
    *flat = -1;
    if (!(rx >= 0 && rx < Dx && ry >= 0 && ry < Dy && rz >= 0 && rz < Dz)) return (REAL)0;
    long f = ((long)rx * Dy + (long)ry) * Dz + (long)rz;
    *flat = f;
    return V[f];
}

typedef struct {
    long flat[8]; /* -1 when the corner is outside the volume */
    REAL w[8];
    REAL dwx[8], dwy[8], dwz[8]; /* d w / d (index coord) */
} FN(tri_t_);

/* aten grid_sampler_3d "bilinear" (trilinear) with padding_mode="zeros":
 * 8 corners around floor(coord), each corner dropped if out of bounds. */
static inline REAL FN(fetch_trilinear_)(const REAL *V, int Dx, int Dy, int Dz, REAL gx, REAL gy,
                                        REAL gz, FN(tri_t_) * rec) {
    REAL fx = (REAL)floor((double)gx), fy = (REAL)floor((double)gy), fz = (REAL)floor((double)gz);
    REAL ax = gx - fx, ay = gy - fy, az = gz - fz; /* fractional parts */
    REAL acc = 0;
    for (int c = 0; c < 8; ++c) {
        int ox = c & 1, oy = (c >> 1) & 1, oz = (c >> 2) & 1;
        REAL wx = ox ? ax : (REAL)1 - ax, wy = oy ? ay : (REAL)1 - ay, wz = oz ? az : (REAL)1 - az;
        double ix = (double)fx + ox, iy = (double)fy + oy, iz = (double)fz + oz;
        int in = ix >= 0 && ix < Dx && iy >= 0 && iy < Dy && iz >= 0 && iz < Dz;
        long f = in ? ((long)ix * Dy + (long)iy) * Dz + (long)iz : -1;
        REAL w = wx * wy * wz;
        if (rec) {
            rec->flat[c] = f;
            rec->w[c] = w;
            rec->dwx[c] = (ox ? (REAL)1 : (REAL)-1) * wy * wz;
            rec->dwy[c] = (oy ? (REAL)1 : (REAL)-1) * wx * wz;
            rec->dwz[c] = (oz ? (REAL)1 : (REAL)-1) * wx * wy;
        }
        if (in) acc += V[f] * w;
    }
    return acc;
}

/* ------------------------------------------------------- Siddon: one ray */

typedef struct {
    REAL alpha;
    int axis;
} FN(cross_t_);

/* _get_alphas (renderers.py:94-113): alpha = (plane - s) / (t - s + eps) for
 * planes i - shift, i = 0..D, per axis; cat; sort.  Each axis' sequence is
 * monotone in i, so torch.sort of the concatenation is realised as a 3-way
 * merge (same values, same order up to ties). */
static int FN(sorted_crossings_)(const REAL s[3], const REAL t[3], const int D[3], REAL shift,
                                 REAL eps, FN(cross_t_) * out, REAL *scratch) {
    int M = D[0] + D[1] + D[2] + 3;
    REAL *seq[3];
    int len[3];
    REAL *p = scratch;
    for (int a = 0; a < 3; ++a) {
        REAL d = t[a] - s[a] + eps; /* renderers.py:104-106 */
        len[a] = D[a] + 1;
        seq[a] = p;
        p += len[a];
        for (int i = 0; i <= D[a]; ++i) seq[a][i] = (((REAL)i - shift) - s[a]) / d;
        if (len[a] > 1 && seq[a][0] > seq[a][len[a] - 1]) { /* descending -> reverse */
            for (int i = 0, j = len[a] - 1; i < j; ++i, --j) {
                REAL tmp = seq[a][i];
                seq[a][i] = seq[a][j];
                seq[a][j] = tmp;
            }
        }
    }
    int h[3] = {0, 0, 0};
    for (int k = 0; k < M; ++k) {
        int best = -1;
        for (int a = 0; a < 3; ++a) {
            if (h[a] >= len[a]) continue;
            if (best < 0 || seq[a][h[a]] < seq[best][h[best]]) best = a;
        }
        out[k].alpha = seq[best][h[best]++];
        out[k].axis = best;
    }
    return M;
}

/* reduce_mode: 0 = sum, 1 = max (renderers.py:175-183) */
static void FN(siddon_ray_)(const REAL *V, const int D[3], const REAL s[3], const REAL t[3], REAL L,
                            REAL shift, REAL eps, int reduce_mode, int interp_bilinear,
                            int align_corners, FN(cross_t_) * cr, REAL *scratch, long *flats,
                            /* outputs */ REAL *out,
                            /* optional backward */ REAL gout, int want_grad, double gs[3],
                            double gt[3], double *gimg, double *gvol, long *n_inside) {
    int M = FN(sorted_crossings_)(s, t, D, shift, eps, cr, scratch);
    REAL d[3] = {t[0] - s[0] + eps, t[1] - s[1] + eps, t[2] - s[2] + eps};
    double acc = 0.0;
    REAL best = 0;
    int kbest = -1;
    REAL *vals = scratch; /* re-use: per-segment sampled value */
    long inside = 0;
    for (int k = 0; k + 1 < M; ++k) {
        REAL a0 = cr[k].alpha, a1 = cr[k + 1].alpha;
        REAL mid = (a0 + a1) / (REAL)2; /* renderers.py:57 */
        /* _get_xyzs renderers.py:146-152 */
        REAL x = s[0] + mid * d[0], y = s[1] + mid * d[1], z = s[2] + mid * d[2];
        REAL gx = FN(grid_coord_)(x, shift, D[0], align_corners);
        REAL gy = FN(grid_coord_)(y, shift, D[1], align_corners);
        REAL gz = FN(grid_coord_)(z, shift, D[2], align_corners);
        long flat = -1;
        REAL v = interp_bilinear ? FN(fetch_trilinear_)(V, D[0], D[1], D[2], gx, gy, gz, NULL)
                                 : FN(fetch_nearest_)(V, D[0], D[1], D[2], gx, gy, gz, &flat);
        if (flat >= 0 && a1 > a0) ++inside;
        vals[k] = v;
        flats[k] = flat;
        REAL term = (L * v) * (a1 - a0); /* renderers.py:166 then :70-71 */
        if (reduce_mode == 0) {
            acc += (double)term;
        } else if (kbest < 0 || term > best) {
            best = term;
            kbest = k;
        }
    }
    *out = reduce_mode == 0 ? (REAL)acc : best;
    if (n_inside) *n_inside = inside;
    if (!want_grad) return;

    /* Autograd of the literal program (nearest mode: grid_sample has zero
     * gradient w.r.t. the coordinates, so pose gradients flow through
     * torch.diff(alphas) only; SortBackward routes them to the crossing that
     * produced each alpha):
     *   d out / d alpha_j = L (V_{j-1} - V_j),  V_{-1} = V_{M-1} = 0
     *   d alpha / d s_a = (alpha - 1) / d_a,    d alpha / d t_a = -alpha / d_a
     *   d out / d img   = sum_k V_k dalpha_k,   d out / d V[voxel_k] = L dalpha_k
     * (max: only the arg-max segment is live). */
    double g = (double)gout;
    double I = 0.0;
    for (int k = 0; k + 1 < M; ++k) {
        if (reduce_mode == 1 && k != kbest) continue;
        double da = (double)cr[k + 1].alpha - (double)cr[k].alpha;
        I += (double)vals[k] * da;
        if (gvol && flats[k] >= 0) {
#pragma omp atomic
            gvol[flats[k]] += g * (double)L * da;
        }
    }
    if (gimg) *gimg = g * I;
    for (int a = 0; a < 3; ++a) gs[a] = gt[a] = 0.0;
    for (int j = 0; j < M; ++j) {
        double vprev = 0.0, vcur = 0.0;
        if (reduce_mode == 0) {
            vprev = j > 0 ? (double)vals[j - 1] : 0.0;
            vcur = j + 1 < M ? (double)vals[j] : 0.0;
        } else {
            if (j == kbest + 1) vprev = (double)vals[kbest];
            if (j == kbest) vcur = (double)vals[kbest];
        }
        double c = g * (double)L * (vprev - vcur);
        if (c == 0.0) continue;
        int a = cr[j].axis;
        double al = (double)cr[j].alpha;
        gs[a] += c * (al - 1.0) / (double)d[a];
        gt[a] += c * (-al) / (double)d[a];
    }
}

/* -------------------------------------------------------------- Siddon API */

/* Siddon.forward (renderers.py:34-76, mask=None branch).
 * src_n: 1 (source broadcast over rays) or N.  img may be NULL (treated as 1).
 * Optional outputs (NULL to skip): g_source (B,N,3) PER RAY (callers sum over
 * rays for a broadcast source), g_target (B,N,3), g_img (B,N), g_volume
 * (Dx*Dy*Dz doubles, accumulated), n_inside (B,N) = number of positive-length
 * segments whose midpoint is inside the volume (SURVEY 8(d) n_vox). */
void FN(oracle_siddon_)(const REAL *V, int Dx, int Dy, int Dz, const REAL *source, int src_n,
                        const REAL *target, const REAL *img, int B, int N, REAL shift, REAL eps,
                        int reduce_mode, int interp_bilinear, int align_corners, REAL *out,
                        const REAL *grad_out, double *g_source, double *g_target, double *g_img,
                        double *g_volume, long *n_inside) {
    const int D[3] = {Dx, Dy, Dz};
    const int M = Dx + Dy + Dz + 3;
    const long R = (long)B * N;
    const int want_grad = grad_out != NULL;
#pragma omp parallel
    {
        FN(cross_t_) *cr = (FN(cross_t_) *)malloc(sizeof(FN(cross_t_)) * (size_t)M);
        REAL *scratch = (REAL *)malloc(sizeof(REAL) * (size_t)(M + 8));
        long *flats = (long *)malloc(sizeof(long) * (size_t)M);
#pragma omp for schedule(dynamic, 64)
        for (long r = 0; r < R; ++r) {
            long b = r / N, n = r % N;
            const REAL *s = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3;
            const REAL *t = target + r * 3;
            REAL L = img ? img[r] : (REAL)1;
            double gs[3], gt[3], gi = 0.0;
            FN(siddon_ray_)(V, D, s, t, L, shift, eps, reduce_mode, interp_bilinear, align_corners,
                            cr, scratch, flats, out + r, want_grad ? grad_out[r] : (REAL)0, want_grad, gs,
                            gt, &gi, g_volume, n_inside ? n_inside + r : NULL);
            if (want_grad) {
                for (int a = 0; a < 3; ++a) {
                    if (g_source) g_source[r * 3 + a] = gs[a];
                    if (g_target) g_target[r * 3 + a] = gt[a];
                }
                if (g_img) g_img[r] = gi;
            }
        }
        free(cr);
        free(scratch);
        free(flats);
    }
}

/* Per-segment terms of one batch, in sorted-alpha order: the (B,N,M-1) tensor
 * the reference holds just before `reduce` (renderers.py:71), together with
 * the nearest-neighbour label of each segment when a mask volume is given
 * (renderers.py:80-84).  Used to check mask_to_channels and callable reducefn. */
void FN(oracle_siddon_segments_)(const REAL *V, const REAL *mask, int Dx, int Dy, int Dz,
                                 const REAL *source, int src_n, const REAL *target, const REAL *img,
                                 int B, int N, REAL shift, REAL eps, REAL *terms, REAL *labels) {
    const int D[3] = {Dx, Dy, Dz};
    const int M = Dx + Dy + Dz + 3;
    const long R = (long)B * N;
#pragma omp parallel
    {
        FN(cross_t_) *cr = (FN(cross_t_) *)malloc(sizeof(FN(cross_t_)) * (size_t)M);
        REAL *scratch = (REAL *)malloc(sizeof(REAL) * (size_t)(M + 8));
#pragma omp for schedule(dynamic, 64)
        for (long r = 0; r < R; ++r) {
            long b = r / N, n = r % N;
            const REAL *s = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3;
            const REAL *t = target + r * 3;
            REAL L = img ? img[r] : (REAL)1;
            FN(sorted_crossings_)(s, t, D, shift, eps, cr, scratch);
            REAL d[3] = {t[0] - s[0] + eps, t[1] - s[1] + eps, t[2] - s[2] + eps};
            for (int k = 0; k + 1 < M; ++k) {
                REAL mid = (cr[k].alpha + cr[k + 1].alpha) / (REAL)2;
                REAL gx = FN(grid_coord_)(s[0] + mid * d[0], shift, Dx, 0);
                REAL gy = FN(grid_coord_)(s[1] + mid * d[1], shift, Dy, 0);
                REAL gz = FN(grid_coord_)(s[2] + mid * d[2], shift, Dz, 0);
                long flat;
                REAL v = FN(fetch_nearest_)(V, Dx, Dy, Dz, gx, gy, gz, &flat);
                terms[r * (M - 1) + k] = (L * v) * (cr[k + 1].alpha - cr[k].alpha);
                if (labels) labels[r * (M - 1) + k] = (mask && flat >= 0) ? mask[flat] : (REAL)0;
            }
        }
        free(cr);
        free(scratch);
    }
}

/* ---------------------------------------------------------- alpha min/max */

/* _get_alpha_minmax (renderers.py:124-140): per ray; note the far plane is
 * dims + 1 - shift (one voxel past the face) exactly as the reference has it. */
void FN(oracle_alpha_minmax_)(const REAL *source, int src_n, const REAL *target, int B, int N,
                              int Dx, int Dy, int Dz, REAL shift, REAL eps, REAL *amin, REAL *amax) {
    const int D[3] = {Dx, Dy, Dz};
    for (long r = 0; r < (long)B * N; ++r) {
        long b = r / N, n = r % N;
        const REAL *s = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3;
        const REAL *t = target + r * 3;
        REAL lo = 0, hi = 0;
        for (int a = 0; a < 3; ++a) {
            REAL sdd = t[a] - s[a] + eps;
            REAL a0 = (((REAL)0 - shift) - s[a]) / sdd;
            REAL a1 = (((REAL)(D[a] + 1) - shift) - s[a]) / sdd;
            REAL mn = a0 < a1 ? a0 : a1, mx = a0 < a1 ? a1 : a0;
            if (a == 0 || mn > lo) lo = mn;
            if (a == 0 || mx < hi) hi = mx;
        }
        amin[r] = lo < (REAL)0 ? (REAL)0 : lo;
        amax[r] = hi > (REAL)1 ? (REAL)1 : hi;
    }
}

/* ------------------------------------------------------------- Trilinear */

/* Trilinear.forward (renderers.py:205-241, mask=None branch) with alphamin /
 * alphamax supplied (the caller reproduces :220-223 with oracle_alpha_minmax +
 * a global min / max).  mode: 1 = trilinear ("bilinear"), 0 = nearest.
 * lin01 = the P values of torch.linspace(0, 1, P): the reference builds them
 * in the DEFAULT dtype (fp32) and only then casts `.to(volume)`
 * (renderers.py:224), and aten's vectorised CPU kernel rounds them slightly
 * differently from the textbook formula, so the table is an input here.
 * Backward outputs follow SURVEY 8(a): per-ray g_source/g_target through the
 * sample positions only, g_img, g_volume, and the two scalars g_alphamin /
 * g_alphamax (summed over all rays). */
void FN(oracle_trilinear_)(const REAL *V, int Dx, int Dy, int Dz, const REAL *source, int src_n,
                           const REAL *target, const REAL *img, int B, int N, REAL shift, REAL eps,
                           int n_points, const float *lin01, REAL alphamin, REAL alphamax,
                           int mode_trilinear,
                           int reduce_mode, int align_corners, REAL *out, const REAL *grad_out,
                           double *g_source, double *g_target, double *g_img, double *g_volume,
                           double *g_alphamin, double *g_alphamax) {
    const long R = (long)B * N;
    const int P = n_points;
    const REAL span = alphamax - alphamin;
    const REAL step = span / (REAL)(P - 1); /* renderers.py:235 */
    double ga_min = 0.0, ga_max = 0.0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : ga_min, ga_max)
    for (long r = 0; r < R; ++r) {
        long b = r / N, n = r % N;
        const REAL *s = source + (b * src_n + (src_n == 1 ? 0 : n)) * 3;
        const REAL *t = target + r * 3;
        REAL L = img ? img[r] : (REAL)1;
        REAL d[3] = {t[0] - s[0] + eps, t[1] - s[1] + eps, t[2] - s[2] + eps};
        double acc = 0.0;
        REAL best = 0;
        int mbest = -1;
        for (int m = 0; m < P; ++m) {
            REAL al = (REAL)lin01[m] * span + alphamin; /* renderers.py:224-225 */
            REAL gx = FN(grid_coord_)(s[0] + al * d[0], shift, Dx, align_corners);
            REAL gy = FN(grid_coord_)(s[1] + al * d[1], shift, Dy, align_corners);
            REAL gz = FN(grid_coord_)(s[2] + al * d[2], shift, Dz, align_corners);
            long flat;
            REAL v = mode_trilinear ? FN(fetch_trilinear_)(V, Dx, Dy, Dz, gx, gy, gz, NULL)
                                    : FN(fetch_nearest_)(V, Dx, Dy, Dz, gx, gy, gz, &flat);
            REAL term = (L * v) * step; /* renderers.py:166 then :236 */
            if (reduce_mode == 0)
                acc += (double)term;
            else if (mbest < 0 || term > best) {
                best = term;
                mbest = m;
            }
        }
        out[r] = reduce_mode == 0 ? (REAL)acc : best;
        if (!grad_out) continue;

        double g = (double)grad_out[r];
        double gs[3] = {0, 0, 0}, gt[3] = {0, 0, 0}, sumT = 0.0;
        /* coordinate scale d(index coord)/d(x): 1 for align_corners=False */
        double cs[3] = {1.0, 1.0, 1.0};
        if (align_corners) {
            cs[0] = (double)(Dx - 1) / Dx;
            cs[1] = (double)(Dy - 1) / Dy;
            cs[2] = (double)(Dz - 1) / Dz;
        }
        for (int m = 0; m < P; ++m) {
            if (reduce_mode == 1 && m != mbest) continue;
            REAL u = (REAL)lin01[m];
            REAL al = u * span + alphamin;
            REAL gx = FN(grid_coord_)(s[0] + al * d[0], shift, Dx, align_corners);
            REAL gy = FN(grid_coord_)(s[1] + al * d[1], shift, Dy, align_corners);
            REAL gz = FN(grid_coord_)(s[2] + al * d[2], shift, Dz, align_corners);
            double T, dT[3] = {0, 0, 0};
            if (mode_trilinear) {
                FN(tri_t_) rec;
                T = (double)FN(fetch_trilinear_)(V, Dx, Dy, Dz, gx, gy, gz, &rec);
                for (int c = 0; c < 8; ++c) {
                    if (rec.flat[c] < 0) continue;
                    double v = (double)V[rec.flat[c]];
                    dT[0] += v * (double)rec.dwx[c] * cs[0];
                    dT[1] += v * (double)rec.dwy[c] * cs[1];
                    dT[2] += v * (double)rec.dwz[c] * cs[2];
                    if (g_volume) {
#pragma omp atomic
                        g_volume[rec.flat[c]] += g * (double)L * (double)step * (double)rec.w[c];
                    }
                }
            } else {
                long flat;
                T = (double)FN(fetch_nearest_)(V, Dx, Dy, Dz, gx, gy, gz, &flat);
                if (g_volume && flat >= 0) {
#pragma omp atomic
                    g_volume[flat] += g * (double)L * (double)step;
                }
            }
            sumT += T;
            double k = g * (double)L * (double)step;
            double ddot = dT[0] * (double)d[0] + dT[1] * (double)d[1] + dT[2] * (double)d[2];
            for (int a = 0; a < 3; ++a) {
                gs[a] += k * (1.0 - (double)al) * dT[a]; /* x = s + al (t - s + eps) */
                gt[a] += k * (double)al * dT[a];
            }
            ga_min += k * (1.0 - (double)u) * ddot; /* through x(alpha_m) */
            ga_max += k * (double)u * ddot;
        }
        /* through step_size = (alphamax - alphamin)/(P-1) */
        ga_min += -g * (double)L * sumT / (double)(P - 1);
        ga_max += g * (double)L * sumT / (double)(P - 1);
        if (g_img) g_img[r] = g * sumT * (double)step;
        for (int a = 0; a < 3; ++a) {
            if (g_source) g_source[r * 3 + a] = gs[a];
            if (g_target) g_target[r * 3 + a] = gt[a];
        }
    }
    if (g_alphamin) *g_alphamin = ga_min;
    if (g_alphamax) *g_alphamax = ga_max;
}

#undef CAT_
#undef CAT
#undef FN
