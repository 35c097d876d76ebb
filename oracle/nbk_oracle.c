/*
 * oracle/nbk_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See nbk_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fPIC -shared -pthread nbk_oracle.c -lm   (oracle/Makefile)
 * -ffp-contract=off + explicit fma() = the arithmetic contract of DESIGN.md: every rounding below is
 * written out, so an independent implementation that follows the same order is bit-identical.
 */
#include "nbk_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FMA(a, b, c) fma((a), (b), (c))

/* ------------------------------------------------------------------------------------------------
 * small vector helpers (operation order is part of the contract)
 * ---------------------------------------------------------------------------------------------- */
static inline double dot3(const double *a, const double *b) {
    return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0]));
}
static inline void sub3(const double *a, const double *b, double *o) {
    o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2];
}
static inline void cross3(const double *a, const double *b, double *o) {
    o[0] = FMA(a[1], b[2], -(a[2] * b[1]));
    o[1] = FMA(a[2], b[0], -(a[0] * b[2]));
    o[2] = FMA(a[0], b[1], -(a[1] * b[0]));
}
/* o = a + s*b */
static inline void axpy3(double s, const double *b, const double *a, double *o) {
    o[0] = FMA(s, b[0], a[0]); o[1] = FMA(s, b[1], a[1]); o[2] = FMA(s, b[2], a[2]);
}
static inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* 3x4 transforms: R row-major [9], t [3] */
typedef struct { double R[9]; double t[3]; } xf_t;

static inline void xf_from12(const double *p, xf_t *x) {
    for (int i = 0; i < 3; ++i) {
        x->R[3 * i + 0] = p[4 * i + 0]; x->R[3 * i + 1] = p[4 * i + 1]; x->R[3 * i + 2] = p[4 * i + 2];
        x->t[i] = p[4 * i + 3];
    }
}
/* o = a * b  (b: rotation Rb, translation tb) */
static inline void xf_mul(const xf_t *a, const double *Rb, const double *tb, xf_t *o) {
    for (int i = 0; i < 3; ++i) {
        const double a0 = a->R[3 * i], a1 = a->R[3 * i + 1], a2 = a->R[3 * i + 2];
        for (int j = 0; j < 3; ++j)
            o->R[3 * i + j] = FMA(a2, Rb[6 + j], FMA(a1, Rb[3 + j], a0 * Rb[j]));
        o->t[i] = FMA(a2, tb[2], FMA(a1, tb[1], FMA(a0, tb[0], a->t[i])));
    }
}

/* ------------------------------------------------------------------------------------------------
 * sincos: Cody-Waite reduction by pi/2 (three 33-bit pieces, fused) + fdlibm minimax kernels.
 * Defined for |x| < 2^31; otherwise (and for NaN) both results are NaN.
 * ---------------------------------------------------------------------------------------------- */
static const double TWO_OVER_PI = 6.36619772367581382433e-01;
static const double PIO2_1 = 1.57079632673412561417e+00;
static const double PIO2_2 = 6.07710050630396597660e-11;
static const double PIO2_3 = 2.02226624871116645580e-21;
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

void orc_sincos(double x, double *s, double *c) {
    if (!(fabs(x) < 2147483648.0)) { *s = NAN; *c = NAN; return; }
    const double k = rint(x * TWO_OVER_PI);
    double r = FMA(-k, PIO2_1, x);
    r = FMA(-k, PIO2_2, r);
    r = FMA(-k, PIO2_3, r);
    const double z = r * r;
    double ps = FMA(z, S6, S5);
    ps = FMA(z, ps, S4); ps = FMA(z, ps, S3); ps = FMA(z, ps, S2); ps = FMA(z, ps, S1);
    const double sr = FMA(r * z, ps, r);
    double pc = FMA(z, C6, C5);
    pc = FMA(z, pc, C4); pc = FMA(z, pc, C3); pc = FMA(z, pc, C2); pc = FMA(z, pc, C1);
    const double cr = FMA(z * z, pc, FMA(-0.5, z, 1.0));
    const int n = (int)((long long)k & 3LL);
    switch (n) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
void orc_sincos_array(const double *x, int64_t n, double *s, double *c) {
    for (int64_t i = 0; i < n; ++i) orc_sincos(x[i], &s[i], &c[i]);
}
void orc_sqrt_div_array(const double *a, const double *b, int64_t n, double *sq, double *dv) {
    for (int64_t i = 0; i < n; ++i) { sq[i] = sqrt(a[i]); dv[i] = a[i] / b[i]; }
}

/* ------------------------------------------------------------------------------------------------
 * kinematics
 * ---------------------------------------------------------------------------------------------- */
/* child frame of joint k given the parent frame and q (helpers.py:43-55 restated with M0/M1/M2) */
static void joint_apply(const orc_model *m, int k, const xf_t *parent, const double *q, xf_t *out) {
    const double *M = m->joint_rot + 27 * k;
    const double *toff = m->joint_trans + 3 * k;
    const double *sl = m->joint_slide + 3 * k;
    const double qk = q[m->joint_qidx[k]];
    double s = 0.0, c = 0.0;
    if (m->joint_type[k] == ORC_REVOLUTE) orc_sincos(qk, &s, &c);
    double L[9], tl[3];
    for (int e = 0; e < 9; ++e) L[e] = FMA(s, M[18 + e], FMA(-c, M[9 + e], M[e]));
    for (int i = 0; i < 3; ++i) tl[i] = FMA(qk, sl[i], toff[i]);
    xf_mul(parent, L, tl, out);
}

static void sweep_path(const orc_model *m, const double *q, const int32_t *path, int32_t path_len, xf_t *T,
                       xf_t *frames /* optional [path_len] */) {
    xf_from12(m->base_pose, T);
    for (int i = 0; i < path_len; ++i) {
        xf_t nxt;
        joint_apply(m, path[i], T, q, &nxt);
        *T = nxt;
        if (frames) frames[i] = nxt;
    }
}

static void xf_to16(const xf_t *x, double *o) {
    for (int i = 0; i < 3; ++i) {
        o[4 * i + 0] = x->R[3 * i]; o[4 * i + 1] = x->R[3 * i + 1]; o[4 * i + 2] = x->R[3 * i + 2];
        o[4 * i + 3] = x->t[i];
    }
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
}
static void xf_from16(const double *p, xf_t *x) {
    for (int i = 0; i < 3; ++i) {
        x->R[3 * i + 0] = p[4 * i + 0]; x->R[3 * i + 1] = p[4 * i + 1]; x->R[3 * i + 2] = p[4 * i + 2];
        x->t[i] = p[4 * i + 3];
    }
}

int orc_fk(const orc_model *m, const double *q, int64_t B, const int32_t *path, int32_t path_len,
           const double *local, const double *local_pose, double *out) {
    xf_t loc;
    xf_from12(local, &loc);
    for (int64_t b = 0; b < B; ++b) {
        xf_t T, E;
        sweep_path(m, q + b * m->n_q, path, path_len, &T, NULL);
        xf_mul(&T, loc.R, loc.t, &E);
        if (local_pose) {
            xf_t lp, E2;
            xf_from16(local_pose + 16 * b, &lp);
            xf_mul(&E, lp.R, lp.t, &E2);
            E = E2;
        }
        xf_to16(&E, out + 16 * b);
    }
    return 0;
}

int orc_jacobian(const orc_model *m, const double *q, int64_t B, const int32_t *path, int32_t path_len,
                 const double *local, int32_t mode, const double *pose, double *out) {
    xf_t loc;
    xf_from12(local, &loc);
    const int nq = m->n_q;
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(path_len > 0 ? path_len : 1));
    if (!frames) return -1;
    for (int64_t b = 0; b < B; ++b) {
        double *J = out + (size_t)b * 6 * nq;
        memset(J, 0, sizeof(double) * 6 * (size_t)nq);
        xf_t T, E;
        sweep_path(m, q + b * nq, path, path_len, &T, frames);
        xf_mul(&T, loc.R, loc.t, &E);
        double pend[3];
        if (mode == 1) {
            xf_t lp, E2;
            xf_from16(pose + 16 * b, &lp);
            xf_mul(&E, lp.R, lp.t, &E2);
            pend[0] = E2.t[0]; pend[1] = E2.t[1]; pend[2] = E2.t[2];
        } else if (mode == 2) {
            pend[0] = pose[16 * b + 3]; pend[1] = pose[16 * b + 7]; pend[2] = pose[16 * b + 11];
        } else {
            pend[0] = E.t[0]; pend[1] = E.t[1]; pend[2] = E.t[2];
        }
        for (int i = 0; i < path_len; ++i) {
            const int k = path[i];
            const double *a = m->joint_axis + 3 * k;
            const xf_t *F = &frames[i];
            double w[3];
            for (int r = 0; r < 3; ++r) w[r] = FMA(F->R[3 * r + 2], a[2], FMA(F->R[3 * r + 1], a[1], F->R[3 * r] * a[0]));
            const int col = m->joint_qidx[k];
            if (m->joint_type[k] == ORC_REVOLUTE) {
                double d[3], v[3];
                sub3(pend, F->t, d);
                cross3(w, d, v);
                J[0 * nq + col] = v[0]; J[1 * nq + col] = v[1]; J[2 * nq + col] = v[2];
                J[3 * nq + col] = w[0]; J[4 * nq + col] = w[1]; J[5 * nq + col] = w[2];
            } else {
                J[0 * nq + col] = w[0]; J[1 * nq + col] = w[1]; J[2 * nq + col] = w[2];
            }
        }
    }
    free(frames);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Levenberg-Marquardt inverse kinematics (numbotics/robots/arm.py:464-552; rot_diff: numbotics/math/spatial.py:207-212)
 * ---------------------------------------------------------------------------------------------- */
static double ik_diff(const xf_t *P, const xf_t *E, double *d) {
    for (int i = 0; i < 3; ++i) d[i] = P->t[i] - E->t[i];
    double R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i][j] = FMA(P->R[3 * i + 2], E->R[3 * j + 2], FMA(P->R[3 * i + 1], E->R[3 * j + 1], P->R[3 * i] * E->R[3 * j]));
    d[3] = 0.5 * (R[2][1] - R[1][2]);
    d[4] = 0.5 * (R[0][2] - R[2][0]);
    d[5] = 0.5 * (R[1][0] - R[0][1]);
    double s = d[0] * d[0];
    for (int i = 1; i < 6; ++i) s = FMA(d[i], d[i], s);
    return sqrt(s);
}

/* pose of the frame and its 6 x n_q Jacobian (rows [v; w], as orc_jacobian mode 0) at q */
static void ik_sweep(const orc_model *m, const double *q, const int32_t *path, int32_t path_len, const xf_t *loc,
                     xf_t *frames, double *J, xf_t *E) {
    xf_t T;
    const int nq = m->n_q;
    sweep_path(m, q, path, path_len, &T, frames);
    xf_mul(&T, loc->R, loc->t, E);
    for (int i = 0; i < path_len; ++i) {
        const int k = path[i];
        const double *a = m->joint_axis + 3 * k;
        const xf_t *F = &frames[i];
        double w[3];
        for (int r = 0; r < 3; ++r) w[r] = FMA(F->R[3 * r + 2], a[2], FMA(F->R[3 * r + 1], a[1], F->R[3 * r] * a[0]));
        const int col = m->joint_qidx[k];
        if (m->joint_type[k] == ORC_REVOLUTE) {
            double d[3], v[3];
            sub3(E->t, F->t, d);
            cross3(w, d, v);
            for (int r = 0; r < 3; ++r) { J[r * nq + col] = v[r]; J[(3 + r) * nq + col] = w[r]; }
        } else {
            for (int r = 0; r < 3; ++r) { J[r * nq + col] = w[r]; J[(3 + r) * nq + col] = 0.0; }
        }
    }
}

static int ik_solve(const double *J, int nq, double lambda, const double *d, double *x) {
    double A[6][6], L[6][6], y[6];
    int ok = 1;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) A[r][c] = 0.0;
    for (int j = 0; j < nq; ++j)
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) A[r][c] = FMA(J[r * nq + j], J[c * nq + j], A[r][c]);
    for (int r = 0; r < 6; ++r) A[r][r] = A[r][r] + lambda;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) {
            double sum = A[j][i];
            for (int k = 0; k < j; ++k) sum = FMA(-L[i][k], L[j][k], sum);
            if (i == j) { if (!(sum > 0.0)) ok = 0; L[i][i] = sqrt(sum); }
            else L[i][j] = sum / L[j][j];
        }
    }
    for (int i = 0; i < 6; ++i) {
        double sum = d[i];
        for (int k = 0; k < i; ++k) sum = FMA(-L[i][k], y[k], sum);
        y[i] = sum / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double sum = y[i];
        for (int k = i + 1; k < 6; ++k) sum = FMA(-L[k][i], x[k], sum);
        x[i] = sum / L[i][i];
    }
    return ok;
}

int orc_ik(const orc_model *m, const double *pose, const double *q0, int64_t B, const int32_t *path, int32_t path_len,
           const double *local, const double *limits, double tol, int32_t max_iter, int32_t max_failures,
           double *q_out, uint8_t *success, double *diff_norm, int32_t *iters) {
    const int nq = m->n_q;
    xf_t loc;
    xf_from12(local, &loc);
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(path_len > 0 ? path_len : 1));
    double *J = (double *)malloc(sizeof(double) * 6 * (size_t)nq);
    if (!frames || !J) { free(frames); free(J); return -1; }
    for (int64_t b = 0; b < B; ++b) {
        double *q = q_out + b * nq;
        memcpy(q, q0 + b * nq, sizeof(double) * (size_t)nq);
        memset(J, 0, sizeof(double) * 6 * (size_t)nq);
        xf_t P, E;
        xf_from12(pose + 16 * b, &P);
        ik_sweep(m, q, path, path_len, &loc, frames, J, &E);
        double d[6], x[6];
        double nrm = ik_diff(&P, &E, d);
        double lambda = 1e-1;
        int fail = 0, used = 0;
        for (int it = 0; it < max_iter; ++it) {
            if (!((nrm > tol) && (fail < max_failures))) break;
            if (!ik_solve(J, nq, lambda, d, x)) { fail = max_failures; continue; }
            for (int j = 0; j < nq; ++j) {
                double acc = 0.0;
                for (int r = 0; r < 6; ++r) acc = FMA(J[r * nq + j], x[r], acc);
                double qj = q[j] + acc;
                if (limits) { if (qj < limits[2 * j]) qj = limits[2 * j]; if (qj > limits[2 * j + 1]) qj = limits[2 * j + 1]; }
                q[j] = qj;
            }
            ik_sweep(m, q, path, path_len, &loc, frames, J, &E);
            const double prev = nrm;
            nrm = ik_diff(&P, &E, d);
            const int grew = nrm > prev;
            lambda = lambda * (grew ? 1.2 : 0.5);
            fail = grew ? fail + 1 : 0;
            used += 1;
        }
        success[b] = (nrm < tol) ? 1 : 0;
        if (diff_norm) diff_norm[b] = nrm;
        if (iters) iters[b] = used;
    }
    free(frames); free(J);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * narrowphase.  Every shape = convex core (+) ball(margin):
 *   sphere   : point,            margin r
 *   capsule  : segment c +- hl u, margin r            (u = local z)
 *   box      : box he - m,       margin m (default 0)
 *   cylinder : cylinder (R - m, hl - m), margin m (default 0)   (axis = local z)
 * signed distance d = dist(coreA, coreB) - mA - mB when the cores are disjoint (exact), and
 * d = -depth(coreA, coreB) - mA - mB when they overlap, depth = minimum overlap over the candidate
 * axis family below (exact MTD for point/segment/box cores, an upper bound when a cylinder core is
 * involved).
 * ---------------------------------------------------------------------------------------------- */
/* K_HULL: convex hull of a vertex list given in the primitive's local frame (MESH shapes: numbotics/utils/shape.py:81-94
 * hands the mesh file to pybullet.createCollisionShape(GEOM_MESH), which -- without the concave-trimesh flag the reference
 * never sets -- builds one convex hull per OBJ object; numbotics/utils/mesh.py:18-37 scales / offsets the vertices first).
 * The local origin is the mean of the hull's vertices (an interior point), rho = the largest vertex norm. */
enum { K_POINT = 0, K_SEG = 1, K_BOX = 2, K_CYL = 3, K_HULL = 4, K_PLANE = 5 };

typedef struct {
    int kind;
    double c[3];
    double ax[3][3]; /* ax[j] = world direction of local axis j (column j of R) */
    double h[3];     /* seg: h[0] = half length; cyl: h[0] = half height; box: half extents (core) */
    double rad;      /* cyl core radius */
    double margin;
    const double *hv; int hn;   /* hull: vertices [hn][3] (local) */
    const double *hp; int hf;   /* hull: face planes [hf][4] = unit outward normal, offset (local): inside n.x <= d */
    double rho;                 /* hull: bounding radius about c */
} core_t;

/* largest vertex norm, rounded like the device's host code: sqrt(max_k fma(z,z,fma(y,y,x*x))) */
static double hull_bound_radius(const double *v, int n) {
    double best = 0.0;
    for (int k = 0; k < n; ++k) {
        const double r2 = FMA(v[3 * k + 2], v[3 * k + 2], FMA(v[3 * k + 1], v[3 * k + 1], v[3 * k] * v[3 * k]));
        if (r2 > best) best = r2;
    }
    return sqrt(best);
}

static void core_from_shape(const orc_model *m, int type, const xf_t *pose, const double *param, core_t *o) {
    o->c[0] = pose->t[0]; o->c[1] = pose->t[1]; o->c[2] = pose->t[2];
    for (int j = 0; j < 3; ++j) { o->ax[j][0] = pose->R[j]; o->ax[j][1] = pose->R[3 + j]; o->ax[j][2] = pose->R[6 + j]; }
    o->h[0] = o->h[1] = o->h[2] = 0.0; o->rad = 0.0; o->margin = 0.0;
    o->hv = NULL; o->hn = 0; o->hp = NULL; o->hf = 0; o->rho = 0.0;
    switch (type) {
        case ORC_HULL: {
            const int h = (int)param[0];
            o->kind = K_HULL; o->margin = param[3];
            o->hv = m->hull_verts + 3 * (size_t)m->hull_vert_begin[h];
            o->hn = m->hull_vert_begin[h + 1] - m->hull_vert_begin[h];
            o->hp = m->hull_planes + 4 * (size_t)m->hull_face_begin[h];
            o->hf = m->hull_face_begin[h + 1] - m->hull_face_begin[h];
            o->rho = hull_bound_radius(o->hv, o->hn);
        } break;
        case ORC_SPHERE: o->kind = K_POINT; o->margin = param[0]; break;
        case ORC_CAPSULE: o->kind = K_SEG; o->margin = param[0]; o->h[0] = param[1]; break;
        case ORC_BOX:
            o->kind = K_BOX; o->margin = param[3];
            o->h[0] = param[0] - param[3]; o->h[1] = param[1] - param[3]; o->h[2] = param[2] - param[3];
            break;
        case ORC_CYLINDER:
            o->kind = K_CYL; o->margin = param[3];
            o->rad = param[0] - param[3]; o->h[0] = param[1] - param[3];
            break;
        default: /* plane: normal in param[0..2], point = pose translation */
            o->kind = K_PLANE; o->ax[2][0] = param[0]; o->ax[2][1] = param[1]; o->ax[2][2] = param[2];
            break;
    }
}

/* support point of a core in direction d */
static void core_support(const core_t *s, const double *d, double *o) {
    switch (s->kind) {
        case K_POINT: o[0] = s->c[0]; o[1] = s->c[1]; o[2] = s->c[2]; break;
        case K_SEG: {
            const double du = dot3(d, s->ax[2]);
            const double sg = du >= 0.0 ? s->h[0] : -s->h[0];
            axpy3(sg, s->ax[2], s->c, o);
        } break;
        case K_CYL: {
            /* radial direction = d minus its axial part, taken twice and WITHOUT assuming a unit axis:
             *   w = (u.u) d - (d.u) u   is perpendicular to u whatever the length of u (a joint axis given to five digits
             *   leaves link rotations orthonormal to 1e-6 only, robots/helpers.py:43-55 uses the axis as given), and the second
             *   pass removes what the rounding of the first leaves when d is almost axial.  With a plain d - (d.u)u an
             *   exactly axial d kept an axial remainder of (1 - u.u)^2 |d| that the normalisation below blew up to O(rad):
             *   a "support point" off the cylinder, and a boolean walk that enclosed the origin of a separated pair
             *   (fuzz seeds 70350, 70503). */
            const double *u = s->ax[2];
            const double du = dot3(d, u);
            const double uu = dot3(u, u);
            const double sg = du >= 0.0 ? s->h[0] : -s->h[0];
            double w[3], t[3];
            t[0] = uu * d[0]; t[1] = uu * d[1]; t[2] = uu * d[2];
            axpy3(-du, u, t, w);
            const double wu = dot3(w, u);
            t[0] = uu * w[0]; t[1] = uu * w[1]; t[2] = uu * w[2];
            axpy3(-wu, u, t, w);
            const double ww = dot3(w, w);
            axpy3(sg, u, s->c, o);
            /* a direction that is axial to 1e-13 has NO radial part worth the name: what the two passes leave is rounding noise
             * of arbitrary direction (axial included), and scaling it to the radius puts the "support point" off the cylinder --
             * EPA asks for exactly such directions (a face normal of A (-) B that IS the cylinder axis).  The cap centre is a
             * support point of an axial direction; the error of its projection is below rad * 1e-13. */
            const double u4 = (uu * uu) * (uu * uu);
            if (ww > (1e-26 * u4) * dot3(d, d)) {
                const double k = s->rad / sqrt(ww);
                axpy3(k, w, o, o);
            }
        } break;
        case K_HULL: {
            /* direction in local coordinates, first maximum over the vertex list, vertex back to the world */
            const double dl[3] = {dot3(d, s->ax[0]), dot3(d, s->ax[1]), dot3(d, s->ax[2])};
            double best = -INFINITY;
            int bi = 0;
            for (int k = 0; k < s->hn; ++k) {
                const double *v = s->hv + 3 * k;
                const double pr = FMA(v[2], dl[2], FMA(v[1], dl[1], v[0] * dl[0]));
                if (pr > best) { best = pr; bi = k; }
            }
            const double *v = s->hv + 3 * bi;
            o[0] = s->c[0]; o[1] = s->c[1]; o[2] = s->c[2];
            for (int j = 0; j < 3; ++j) axpy3(v[j], s->ax[j], o, o);
        } break;
        default: { /* box */
            o[0] = s->c[0]; o[1] = s->c[1]; o[2] = s->c[2];
            for (int j = 0; j < 3; ++j) {
                const double dj = dot3(d, s->ax[j]);
                const double sj = dj >= 0.0 ? s->h[j] : -s->h[j];
                axpy3(sj, s->ax[j], o, o);
            }
        } break;
    }
}

/* half-width of a (centrally symmetric) core along unit direction n */
static double core_halfwidth(const core_t *s, const double *n) {
    switch (s->kind) {
        case K_POINT: return 0.0;
        case K_SEG: return s->h[0] * fabs(dot3(n, s->ax[2]));
        case K_CYL: {
            const double nu = dot3(n, s->ax[2]);
            const double r2 = FMA(-nu, nu, 1.0);
            return FMA(s->rad, sqrt(r2 > 0.0 ? r2 : 0.0), s->h[0] * fabs(nu));
        }
        default:
            return FMA(s->h[2], fabs(dot3(n, s->ax[2])), FMA(s->h[1], fabs(dot3(n, s->ax[1])), s->h[0] * fabs(dot3(n, s->ax[0]))));
    }
}

/* extents of a core along unit direction n, measured from its centre c: the core spans [-neg, +pos].
 * Symmetric kinds: neg = pos = halfwidth; hull: pos = max_k dl.v_k, neg = -min_k dl.v_k (dl = n in local coordinates). */
static void core_extents(const core_t *s, const double *n, double *neg, double *pos) {
    if (s->kind != K_HULL) { const double hw = core_halfwidth(s, n); *neg = hw; *pos = hw; return; }
    const double dl[3] = {dot3(n, s->ax[0]), dot3(n, s->ax[1]), dot3(n, s->ax[2])};
    double hi = -INFINITY, lo = INFINITY;
    for (int k = 0; k < s->hn; ++k) {
        const double *v = s->hv + 3 * k;
        const double pr = FMA(v[2], dl[2], FMA(v[1], dl[1], v[0] * dl[0]));
        if (pr > hi) hi = pr;
        if (pr < lo) lo = pr;
    }
    *pos = hi; *neg = -lo;
}

/* ---- GJK on cores -------------------------------------------------------------------------- */
#define GJK_MAXIT 64
static const double GJK_EPS_REL = 1e-10;
static const double GJK_TINY2 = 1e-30;

typedef struct { double y[4][3], a[4][3], b[4][3]; double lam[4]; int n; } simplex_t;

static void sx_keep(simplex_t *s, const int *idx, const double *lam, int n) {
    simplex_t t = *s;
    for (int i = 0; i < n; ++i) {
        memcpy(s->y[i], t.y[idx[i]], sizeof(double) * 3);
        memcpy(s->a[i], t.a[idx[i]], sizeof(double) * 3);
        memcpy(s->b[i], t.b[idx[i]], sizeof(double) * 3);
        s->lam[i] = lam[i];
    }
    s->n = n;
}

/* closest point to the origin on segment (i0,i1); writes v, returns kept count/idx/lam */
static int closest_seg(const simplex_t *s, int i0, int i1, double *v, int *idx, double *lam) {
    const double *A = s->y[i0], *Bp = s->y[i1];
    double ab[3];
    sub3(Bp, A, ab);
    const double t = -dot3(A, ab);
    if (t <= 0.0) { v[0] = A[0]; v[1] = A[1]; v[2] = A[2]; idx[0] = i0; lam[0] = 1.0; return 1; }
    const double den = dot3(ab, ab);
    if (t >= den) { v[0] = Bp[0]; v[1] = Bp[1]; v[2] = Bp[2]; idx[0] = i1; lam[0] = 1.0; return 1; }
    const double tt = t / den;
    axpy3(tt, ab, A, v);
    idx[0] = i0; idx[1] = i1; lam[0] = 1.0 - tt; lam[1] = tt;
    return 2;
}

/* closest point to the origin on triangle (i0,i1,i2) -- Voronoi-region walk */
static int closest_tri(const simplex_t *s, int i0, int i1, int i2, double *v, int *idx, double *lam) {
    const double *A = s->y[i0], *Bp = s->y[i1], *Cp = s->y[i2];
    double ab[3], ac[3];
    sub3(Bp, A, ab); sub3(Cp, A, ac);
    const double d1 = -dot3(ab, A), d2 = -dot3(ac, A);
    if (d1 <= 0.0 && d2 <= 0.0) { memcpy(v, A, 24); idx[0] = i0; lam[0] = 1.0; return 1; }
    const double d3 = -dot3(ab, Bp), d4 = -dot3(ac, Bp);
    if (d3 >= 0.0 && d4 <= d3) { memcpy(v, Bp, 24); idx[0] = i1; lam[0] = 1.0; return 1; }
    const double vc = FMA(d1, d4, -(d3 * d2));
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double t = d1 / (d1 - d3);
        axpy3(t, ab, A, v);
        idx[0] = i0; idx[1] = i1; lam[0] = 1.0 - t; lam[1] = t;
        return 2;
    }
    const double d5 = -dot3(ab, Cp), d6 = -dot3(ac, Cp);
    if (d6 >= 0.0 && d5 <= d6) { memcpy(v, Cp, 24); idx[0] = i2; lam[0] = 1.0; return 1; }
    const double vb = FMA(d5, d2, -(d1 * d6));
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double t = d2 / (d2 - d6);
        axpy3(t, ac, A, v);
        idx[0] = i0; idx[1] = i2; lam[0] = 1.0 - t; lam[1] = t;
        return 2;
    }
    const double va = FMA(d3, d6, -(d5 * d4));
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        double bc[3];
        sub3(Cp, Bp, bc);
        axpy3(t, bc, Bp, v);
        idx[0] = i1; idx[1] = i2; lam[0] = 1.0 - t; lam[1] = t;
        return 2;
    }
    const double den = 1.0 / (va + vb + vc);
    const double tv = vb * den, tw = vc * den;
    double tmp[3];
    axpy3(tv, ab, A, tmp);
    axpy3(tw, ac, tmp, v);
    idx[0] = i0; idx[1] = i1; idx[2] = i2; lam[0] = (1.0 - tv) - tw; lam[1] = tv; lam[2] = tw;
    return 3;
}

/* origin outside the plane of (a,b,c) on the side away from d?  flat tetrahedra count as outside */
static int outside_face(const double *a, const double *b, const double *c, const double *d) {
    double ab[3], ac[3], ad[3], n[3];
    sub3(b, a, ab); sub3(c, a, ac); sub3(d, a, ad);
    cross3(ab, ac, n);
    const double sp = -dot3(a, n);
    const double sd = dot3(ad, n);
    return (sp * sd < 0.0) || (sd == 0.0);
}

/* returns 1 if the origin is inside the tetrahedron (overlap), else reduces the simplex */
static int closest_tet(simplex_t *s, double *v) {
    static const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
    double best = INFINITY;
    int bidx[3] = {0, 0, 0}, bn = 0;
    double blam[3] = {0, 0, 0}, bv[3] = {0, 0, 0};
    int any = 0;
    for (int f = 0; f < 4; ++f) {
        if (!outside_face(s->y[F[f][0]], s->y[F[f][1]], s->y[F[f][2]], s->y[F[f][3]])) continue;
        any = 1;
        double cv[3], clam[3];
        int cidx[3];
        const int cn = closest_tri(s, F[f][0], F[f][1], F[f][2], cv, cidx, clam);
        const double dd = dot3(cv, cv);
        if (dd < best) {
            best = dd; bn = cn;
            for (int i = 0; i < cn; ++i) { bidx[i] = cidx[i]; blam[i] = clam[i]; }
            bv[0] = cv[0]; bv[1] = cv[1]; bv[2] = cv[2];
        }
    }
    if (!any) return 1;
    v[0] = bv[0]; v[1] = bv[1]; v[2] = bv[2];
    sx_keep(s, bidx, blam, bn);
    return 0;
}

/* one simplex update shared by the distance and the predicate loops.
 * returns 0 = advanced (v, vv_prev updated), 1 = origin enclosed / touching, 2 = no progress (simplex restored) */
static int gjk_advance(simplex_t *sx, const double *w, const double *sa, const double *sb, double *v, double *vv_prev) {
    const int k = sx->n;
    memcpy(sx->y[k], w, 24); memcpy(sx->a[k], sa, 24); memcpy(sx->b[k], sb, 24);
    sx->lam[k] = 0.0;
    sx->n = k + 1;
    simplex_t saved = *sx;
    double nvv[3];
    int idx[3];
    double lam[3];
    int enclosed = 0;
    if (sx->n == 1) {
        memcpy(nvv, sx->y[0], 24); sx->lam[0] = 1.0;
    } else if (sx->n == 2) {
        const int n = closest_seg(sx, 0, 1, nvv, idx, lam);
        sx_keep(sx, idx, lam, n);
    } else if (sx->n == 3) {
        const int n = closest_tri(sx, 0, 1, 2, nvv, idx, lam);
        sx_keep(sx, idx, lam, n);
    } else {
        enclosed = closest_tet(sx, nvv);
    }
    if (enclosed) { *sx = saved; sx->n = k; return 1; }
    const double nn = dot3(nvv, nvv);
    if (nn <= GJK_TINY2) { *sx = saved; sx->n = k; return 1; }
    if (nn >= *vv_prev) { *sx = saved; sx->n = k; return 2; }
    *vv_prev = nn;
    v[0] = nvv[0]; v[1] = nvv[1]; v[2] = nvv[2];
    return 0;
}

/* GJK distance: returns 1 if the cores overlap (or touch), 0 if separated with v the closest vector
 * (from B to A) and pa/pb witness points.  `sep` records a proven separating plane (v.w > 0): once set,
 * a later "enclosed" verdict is numerical breakdown of a degenerate simplex and ends the iteration
 * instead.  Convergence uses the best lower bound seen so far: |v|^2 - max_k (v_k.w_k)^2/|v_k|^2 <= eps |v|^2. */
static int gjk_cores(const core_t *A, const core_t *Bc, double *pa, double *pb, double *vout, int *iters_out) {
    simplex_t sx;
    sx.n = 0;
    double v[3];
    sub3(A->c, Bc->c, v);
    if (dot3(v, v) == 0.0) { v[0] = 1.0; v[1] = 0.0; v[2] = 0.0; }
    double vv_prev = INFINITY, lb2 = 0.0;
    int overlap = 0, sep = 0, it = 0;
    for (it = 0; it < GJK_MAXIT; ++it) {
        double nv[3] = {-v[0], -v[1], -v[2]};
        double sa[3], sb[3], w[3];
        core_support(A, nv, sa);
        core_support(Bc, v, sb);
        sub3(sa, sb, w);
        const double vv = dot3(v, v);
        const double vw = dot3(v, w);
        if (vw > 0.0) {
            sep = 1;
            const double l2 = (vw * vw) / vv;
            if (l2 > lb2) lb2 = l2;
        }
        if (sx.n > 0 && (vv - lb2) <= GJK_EPS_REL * vv) break;
        int dup = 0;
        for (int i = 0; i < sx.n; ++i)
            if (sx.y[i][0] == w[0] && sx.y[i][1] == w[1] && sx.y[i][2] == w[2]) dup = 1;
        if (dup) break;
        const int st = gjk_advance(&sx, w, sa, sb, v, &vv_prev);
        if (st == 1) { if (!sep) overlap = 1; ++it; break; }
        if (st == 2) { ++it; break; }
    }
    if (iters_out) *iters_out = it;
    if (overlap) return 1;
    vout[0] = v[0]; vout[1] = v[1]; vout[2] = v[2];
    pa[0] = pa[1] = pa[2] = 0.0; pb[0] = pb[1] = pb[2] = 0.0;
    for (int i = 0; i < sx.n; ++i) {
        axpy3(sx.lam[i], sx.a[i], pa, pa);
        axpy3(sx.lam[i], sx.b[i], pb, pb);
    }
    return 0;
}

static double overlap_depth(const core_t *A, const core_t *Bc, double *normal);
static double overlap_depth_exact(const core_t *A, const core_t *Bc, double *normal);
static int overlap_deeper_than(const core_t *A, const core_t *Bc, double x);

/* GJK predicate: is dist(coreA, coreB) < tc ?  Same iteration as gjk_cores, but it stops as soon as the
 * support-plane lower bound reaches tc (free) or the simplex point drops below tc (colliding). */
static __thread long long g_pred_hist[80];       /* iterations of gjk_collides (diagnostic) */
static int gjk_collides_it(const core_t *A, const core_t *Bc, double tc, int *iters);
static int gjk_collides(const core_t *A, const core_t *Bc, double tc) {
    int it = 0;
    const int r = gjk_collides_it(A, Bc, tc, &it);
    g_pred_hist[it < 79 ? it : 79] += 1;
    return r;
}
static int gjk_collides_it(const core_t *A, const core_t *Bc, double tc, int *iters) {
    simplex_t sx;
    sx.n = 0;
    double v[3];
    sub3(A->c, Bc->c, v);
    if (dot3(v, v) == 0.0) { v[0] = 1.0; v[1] = 0.0; v[2] = 0.0; }
    double vv_prev = INFINITY, lb2 = 0.0;
    const double tc2 = tc * tc;
    int sep = 0;
    for (int it = 0; it < GJK_MAXIT; ++it) {
        *iters = it + 1;
        double nv[3] = {-v[0], -v[1], -v[2]};
        double sa[3], sb[3], w[3];
        core_support(A, nv, sa);
        core_support(Bc, v, sb);
        sub3(sa, sb, w);
        const double vv = dot3(v, v);
        const double vw = dot3(v, w);
        if (vw > 0.0) {
            sep = 1;
            if (tc <= 0.0) return 0;
            if (vw * vw >= tc2 * vv) return 0;
            const double l2 = (vw * vw) / vv;
            if (l2 > lb2) lb2 = l2;
        }
        if (sx.n > 0 && (vv - lb2) <= GJK_EPS_REL * vv) break;
        int dup = 0;
        for (int i = 0; i < sx.n; ++i)
            if (sx.y[i][0] == w[0] && sx.y[i][1] == w[1] && sx.y[i][2] == w[2]) dup = 1;
        if (dup) break;
        const int st = gjk_advance(&sx, w, sa, sb, v, &vv_prev);
        if (st == 1) {
            if (sep) break;
            if (tc >= 0.0) return 1;
            return overlap_deeper_than(A, Bc, -tc);
        }
        if (st == 2) break;
        if (tc > 0.0 && vv_prev < tc2) return 1;
    }
    return sqrt(dot3(v, v)) < tc;
}

/* ---- overlap depth over a candidate axis family --------------------------------------------- */
static void try_axis(const core_t *A, const core_t *Bc, const double *delta, const double *n_in, double *best,
                     double *bn) {
    const double nn = dot3(n_in, n_in);
    if (!(nn > 1e-24)) return;
    const double inv = 1.0 / sqrt(nn);
    double n[3] = {n_in[0] * inv, n_in[1] * inv, n_in[2] * inv};
    const double proj = dot3(n, delta); /* delta = cA - cB */
    if (A->kind == K_HULL || Bc->kind == K_HULL) {
        /* not centrally symmetric: A spans [-aN, aP], B spans [-bN, bP] about their centres along n.  Pushing A along +n
         * separates after tp = (aN + bP) - proj, along -n after tm = (aP + bN) + proj; the smaller one is the overlap. */
        double aN, aP, bN, bP;
        core_extents(A, n, &aN, &aP);
        core_extents(Bc, n, &bN, &bP);
        const double tp = (aN + bP) - proj, tm = (aP + bN) + proj;
        const double ov = tp <= tm ? tp : tm;
        if (ov < *best) {
            *best = ov;
            const double sg = tp <= tm ? 1.0 : -1.0;
            bn[0] = sg * n[0]; bn[1] = sg * n[1]; bn[2] = sg * n[2];
        }
        return;
    }
    const double ov = (core_halfwidth(A, n) + core_halfwidth(Bc, n)) - fabs(proj);
    if (ov < *best) {
        *best = ov;
        const double sg = proj >= 0.0 ? 1.0 : -1.0; /* normal from B to A */
        bn[0] = sg * n[0]; bn[1] = sg * n[1]; bn[2] = sg * n[2];
    }
}

/* world direction of face f of a hull core */
static void hull_face_normal(const core_t *s, int f, double *n) {
    const double *pl = s->hp + 4 * f;
    n[0] = 0.0; n[1] = 0.0; n[2] = 0.0;
    for (int j = 0; j < 3; ++j) axpy3(pl[j], s->ax[j], n, n);
}

static int core_axes(const core_t *s, const double **ax) {
    if (s->kind == K_BOX) { ax[0] = s->ax[0]; ax[1] = s->ax[1]; ax[2] = s->ax[2]; return 3; }
    if (s->kind == K_SEG || s->kind == K_CYL) { ax[0] = s->ax[2]; return 1; }
    return 0;
}

static double overlap_depth(const core_t *A, const core_t *Bc, double *normal) {
    double delta[3];
    sub3(A->c, Bc->c, delta);
    double best = INFINITY;
    normal[0] = 1.0; normal[1] = 0.0; normal[2] = 0.0;
    const double *aa[3], *ba[3];
    const int na = core_axes(A, aa), nb = core_axes(Bc, ba);
    for (int i = 0; i < na; ++i) try_axis(A, Bc, delta, aa[i], &best, normal);
    for (int j = 0; j < nb; ++j) try_axis(A, Bc, delta, ba[j], &best, normal);
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            double cr[3];
            cross3(aa[i], ba[j], cr);
            try_axis(A, Bc, delta, cr, &best, normal);
        }
    /* face normals of hull cores (exact depth for a point inside a hull; an upper bound otherwise: edge-edge axes are
     * not tried) */
    if (A->kind == K_HULL) for (int f = 0; f < A->hf; ++f) { double fn[3]; hull_face_normal(A, f, fn); try_axis(A, Bc, delta, fn, &best, normal); }
    if (Bc->kind == K_HULL) for (int f = 0; f < Bc->hf; ++f) { double fn[3]; hull_face_normal(Bc, f, fn); try_axis(A, Bc, delta, fn, &best, normal); }
    /* radial directions of cylinder cores, and the centre line */
    if (A->kind == K_CYL) { double r[3]; axpy3(-dot3(delta, A->ax[2]), A->ax[2], delta, r); try_axis(A, Bc, delta, r, &best, normal); }
    if (Bc->kind == K_CYL) { double r[3]; axpy3(-dot3(delta, Bc->ax[2]), Bc->ax[2], delta, r); try_axis(A, Bc, delta, r, &best, normal); }
    try_axis(A, Bc, delta, delta, &best, normal);
    if (best == INFINITY) best = 0.0; /* two coincident points */
    return best;
}

/* ---- closed forms for point / segment cores -------------------------------------------------- */
static void seg_seg_closest(const core_t *A, const core_t *Bc, double *pa, double *pb) {
    const double *ua = A->ax[2], *ub = Bc->ax[2];
    const double ha = A->h[0], hb = Bc->h[0];
    double r[3];
    sub3(A->c, Bc->c, r);
    const double b = dot3(ua, ub), c = dot3(ua, r), f = dot3(ub, r);
    const double den = FMA(-b, b, 1.0);
    double s = 0.0;
    if (den > 1e-14) s = clampd(FMA(b, f, -c) / den, -ha, ha);
    double t = FMA(b, s, f);
    if (t < -hb) { t = -hb; s = clampd(FMA(b, t, -c), -ha, ha); }
    else if (t > hb) { t = hb; s = clampd(FMA(b, t, -c), -ha, ha); }
    axpy3(s, ua, A->c, pa);
    axpy3(t, ub, Bc->c, pb);
}

/* closest points between two cores of kind point/segment */
static void ps_closest(const core_t *A, const core_t *Bc, double *pa, double *pb) {
    if (A->kind == K_POINT && Bc->kind == K_POINT) {
        memcpy(pa, A->c, 24); memcpy(pb, Bc->c, 24);
    } else if (A->kind == K_POINT) {
        double d[3];
        sub3(A->c, Bc->c, d);
        const double t = clampd(dot3(d, Bc->ax[2]), -Bc->h[0], Bc->h[0]);
        memcpy(pa, A->c, 24);
        axpy3(t, Bc->ax[2], Bc->c, pb);
    } else if (Bc->kind == K_POINT) {
        double d[3];
        sub3(Bc->c, A->c, d);
        const double t = clampd(dot3(d, A->ax[2]), -A->h[0], A->h[0]);
        axpy3(t, A->ax[2], A->c, pa);
        memcpy(pb, Bc->c, 24);
    } else {
        seg_seg_closest(A, Bc, pa, pb);
    }
}

/* point core P against box / cylinder core S: signed core distance, closest surface point, outward normal */
static double point_solid(const double *p, const core_t *S, double *cp, double *nrm) {
    double d[3];
    sub3(p, S->c, d);
    if (S->kind == K_BOX) {
        double x[3], qx[3];
        int outside = 0;
        for (int j = 0; j < 3; ++j) {
            x[j] = dot3(d, S->ax[j]);
            qx[j] = clampd(x[j], -S->h[j], S->h[j]);
            if (fabs(x[j]) > S->h[j]) outside = 1;
        }
        if (outside) {
            memcpy(cp, S->c, 24);
            for (int j = 0; j < 3; ++j) axpy3(qx[j], S->ax[j], cp, cp);
            double e[3];
            sub3(p, cp, e);
            const double dist = sqrt(dot3(e, e));
            const double inv = 1.0 / dist;
            nrm[0] = e[0] * inv; nrm[1] = e[1] * inv; nrm[2] = e[2] * inv;
            return dist;
        }
        int jm = 0;
        double best = S->h[0] - fabs(x[0]);
        for (int j = 1; j < 3; ++j) { const double g = S->h[j] - fabs(x[j]); if (g < best) { best = g; jm = j; } }
        const double sg = x[jm] >= 0.0 ? 1.0 : -1.0;
        nrm[0] = sg * S->ax[jm][0]; nrm[1] = sg * S->ax[jm][1]; nrm[2] = sg * S->ax[jm][2];
        axpy3(best, nrm, p, cp);
        return -best;
    }
    /* cylinder */
    const double *u = S->ax[2];
    const double z = dot3(d, u);
    double w[3];
    axpy3(-z, u, d, w);
    const double rho = sqrt(dot3(w, w));
    const double dz = fabs(z) - S->h[0], dr = rho - S->rad;
    const double sz = z >= 0.0 ? 1.0 : -1.0;
    double rdir[3];
    if (rho > 0.0) { const double inv = 1.0 / rho; rdir[0] = w[0] * inv; rdir[1] = w[1] * inv; rdir[2] = w[2] * inv; }
    else { /* on the axis: any radial direction; take the box-like axis 0 of the pose */
        rdir[0] = S->ax[0][0]; rdir[1] = S->ax[0][1]; rdir[2] = S->ax[0][2];
    }
    if (dz <= 0.0 && dr <= 0.0) {
        if (dr > dz) { nrm[0] = rdir[0]; nrm[1] = rdir[1]; nrm[2] = rdir[2]; axpy3(-dr, nrm, p, cp); return dr; }
        nrm[0] = sz * u[0]; nrm[1] = sz * u[1]; nrm[2] = sz * u[2];
        axpy3(-dz, nrm, p, cp);
        return dz;
    }
    const double zc = clampd(z, -S->h[0], S->h[0]);
    const double rc = rho < S->rad ? rho : S->rad;
    double tmp[3];
    axpy3(zc, u, S->c, tmp);
    axpy3(rc, rdir, tmp, cp);
    double e[3];
    sub3(p, cp, e);
    const double dist = sqrt(dot3(e, e));
    const double inv = 1.0 / dist;
    nrm[0] = e[0] * inv; nrm[1] = e[1] * inv; nrm[2] = e[2] * inv;
    return dist;
}

/* signed distance between two shapes given as cores; witness = pa(3) pb(3) n(3), n from B to A */
static double cores_distance(const core_t *A, const core_t *Bc, double *wit, int *iters) {
    double pa[3], pb[3], n[3];
    double dc;
    if (iters) *iters = 0;
    if (Bc->kind == K_PLANE) {
        const double *nn = Bc->ax[2];
        double d[3];
        sub3(A->c, Bc->c, d);
        const double hc = dot3(d, nn);
        double hw, hpos;
        core_extents(A, nn, &hw, &hpos);       /* how far the core reaches below its centre */
        const double dist = (hc - hw) - A->margin;
        if (wit) {
            double neg[3] = {-nn[0], -nn[1], -nn[2]};
            core_support(A, neg, pa);
            axpy3(-A->margin, nn, pa, pa);
            axpy3(-dist, nn, pa, pb);
            memcpy(wit, pa, 24); memcpy(wit + 3, pb, 24); memcpy(wit + 6, nn, 24);
        }
        return dist;
    }
    const int a_ps = (A->kind == K_POINT || A->kind == K_SEG), b_ps = (Bc->kind == K_POINT || Bc->kind == K_SEG);
    if (a_ps && b_ps) {
        ps_closest(A, Bc, pa, pb);
        double e[3];
        sub3(pa, pb, e);
        dc = sqrt(dot3(e, e));
        if (dc > 0.0) { const double inv = 1.0 / dc; n[0] = e[0] * inv; n[1] = e[1] * inv; n[2] = e[2] * inv; }
        else {
            double cr[3];
            cross3(A->ax[2], Bc->ax[2], cr);
            const double cc = dot3(cr, cr);
            if (A->kind == K_SEG && Bc->kind == K_SEG && cc > 1e-24) { const double inv = 1.0 / sqrt(cc); n[0] = cr[0] * inv; n[1] = cr[1] * inv; n[2] = cr[2] * inv; }
            else { n[0] = 1.0; n[1] = 0.0; n[2] = 0.0; }
        }
    } else if (A->kind == K_POINT && Bc->kind != K_HULL) {
        double nb[3];
        dc = point_solid(A->c, Bc, pb, nb);
        memcpy(pa, A->c, 24);
        n[0] = nb[0]; n[1] = nb[1]; n[2] = nb[2];
    } else if (Bc->kind == K_POINT && A->kind != K_HULL) {
        double na[3];
        dc = point_solid(Bc->c, A, pa, na);
        memcpy(pb, Bc->c, 24);
        n[0] = -na[0]; n[1] = -na[1]; n[2] = -na[2];
    } else {
        double v[3];
        const int ov = gjk_cores(A, Bc, pa, pb, v, iters);
        if (!ov) {
            /* the distance is the norm of GJK's own closest vector; pa/pb only serve as witnesses */
            dc = sqrt(dot3(v, v));
            const double inv = 1.0 / dc;
            n[0] = v[0] * inv; n[1] = v[1] * inv; n[2] = v[2] * inv;
        } else {
            const double depth = overlap_depth_exact(A, Bc, n);
            dc = -depth;
            double neg[3] = {-n[0], -n[1], -n[2]};
            core_support(A, neg, pa);      /* deepest point of A along the normal */
            axpy3(dc, n, pa, pb);          /* pb = pa - depth*n */
        }
    }
    const double dist = (dc - A->margin) - Bc->margin;
    if (wit) {
        double wa[3], wb[3];
        axpy3(-A->margin, n, pa, wa);
        axpy3(Bc->margin, n, pb, wb);
        memcpy(wit, wa, 24); memcpy(wit + 3, wb, 24); memcpy(wit + 6, n, 24);
    }
    return dist;
}

/* ---- boolean GJK: do two cores intersect?  (the predicate for tc == 0) ----------------------------------------
 * No closest points, only a walk of the simplex towards the origin of the Minkowski difference M = A (-) B:
 * a support point a with a.d < 0 proves a separating direction (free); a tetrahedron around the origin, or the
 * origin on the simplex, is an intersection.  Undecided after the cap (grazing contact of curved cores) counts
 * as touching.  About five times cheaper per iteration than the distance iteration, which matters because most
 * items that reach this stage are penetrations (>= 4 iterations). */
#define GJKB_MAXIT 32
typedef struct { double p[4][3]; int n; double d[3]; int it; } gjkb_t;

static void mink_support(const core_t *A, const core_t *Bc, const double *d, double *w) {
    double nd[3] = {-d[0], -d[1], -d[2]}, sa[3], sb[3];
    core_support(A, d, sa);
    core_support(Bc, nd, sb);
    sub3(sa, sb, w);
}
static void tri_prod(const double *x, const double *y, double *o) { /* (x cross y) cross x */
    double t[3];
    cross3(x, y, t);
    cross3(t, x, o);
}
/* triangle (c oldest, b, a newest): new simplex and direction.  keeps points in g->p[0..n) oldest first */
static void gjkb_triangle(gjkb_t *g, const double *c, const double *b, const double *a) {
    double ab[3], ac[3], ao[3] = {-a[0], -a[1], -a[2]}, abc[3], t[3];
    sub3(b, a, ab); sub3(c, a, ac);
    cross3(ab, ac, abc);
    cross3(abc, ac, t);
    int star = 0;
    if (dot3(t, ao) > 0.0) {
        if (dot3(ac, ao) > 0.0) {
            double cc[3] = {c[0], c[1], c[2]}, aa[3] = {a[0], a[1], a[2]};
            memcpy(g->p[0], cc, 24); memcpy(g->p[1], aa, 24); g->n = 2;
            tri_prod(ac, ao, g->d);
            return;
        }
        star = 1;
    } else {
        cross3(ab, abc, t);
        if (dot3(t, ao) > 0.0) star = 1;
    }
    if (star) {
        double bb[3] = {b[0], b[1], b[2]}, aa[3] = {a[0], a[1], a[2]};
        if (dot3(ab, ao) > 0.0) { memcpy(g->p[0], bb, 24); memcpy(g->p[1], aa, 24); g->n = 2; tri_prod(ab, ao, g->d); }
        else { memcpy(g->p[0], aa, 24); g->n = 1; g->d[0] = ao[0]; g->d[1] = ao[1]; g->d[2] = ao[2]; }
        return;
    }
    {
        double cc[3] = {c[0], c[1], c[2]}, bb[3] = {b[0], b[1], b[2]}, aa[3] = {a[0], a[1], a[2]};
        if (dot3(abc, ao) > 0.0) {
            memcpy(g->p[0], cc, 24); memcpy(g->p[1], bb, 24); memcpy(g->p[2], aa, 24);
            g->d[0] = abc[0]; g->d[1] = abc[1]; g->d[2] = abc[2];
        } else {
            memcpy(g->p[0], bb, 24); memcpy(g->p[1], cc, 24); memcpy(g->p[2], aa, 24);
            g->d[0] = -abc[0]; g->d[1] = -abc[1]; g->d[2] = -abc[2];
        }
        g->n = 3;
    }
}

static void gjkb_init(gjkb_t *g, const core_t *A, const core_t *Bc) {
    sub3(A->c, Bc->c, g->d);
    if (dot3(g->d, g->d) == 0.0) { g->d[0] = 1.0; g->d[1] = 0.0; g->d[2] = 0.0; }
    g->n = 0;
    g->it = 0;
}
/* one iteration: 0 = continue, 1 = free, 2 = intersecting */
/* infl (tc > 0): core A is inflated by a ball of radius tc -- its support point moves by tc d/|d| -- so the same walk decides
 * dist(A, B) < tc.  A rounded shape can need many steps to separate from a near-tangent partner (cylinder pairs a few 1e-4
 * apart: up to 38 steps seen in 1e6 configurations of the benchmark scene, where the sharp-shape walk needs 8); with the cap at
 * 64 nothing was left undecided there, with 20 some 10-2000 pairs per 1e6 configurations.  An undecided walk (cap reached, or
 * the origin on the simplex) returns 3 and cores_collide falls back to gjk_collides, so the predicate stays exact. */
#define GJKB_INFL_MAXIT 64
static int gjkb_step(gjkb_t *g, const core_t *A, const core_t *Bc, double tc) {
    const int infl = tc > 0.0;
    if (g->it >= (infl ? GJKB_INFL_MAXIT : GJKB_MAXIT)) return infl ? 3 : 2;
    double a[3];
    mink_support(A, Bc, g->d, a);
    if (infl) {
        const double k = tc / sqrt(dot3(g->d, g->d));
        axpy3(k, g->d, a, a);
    }
    if (dot3(a, g->d) < 0.0) return 1;
    g->it += 1;
    if (g->n == 0) {
        memcpy(g->p[0], a, 24); g->n = 1;
        g->d[0] = -a[0]; g->d[1] = -a[1]; g->d[2] = -a[2];
    } else if (g->n == 1) {
        double b[3], ab[3], ao[3] = {-a[0], -a[1], -a[2]};
        memcpy(b, g->p[0], 24);
        sub3(b, a, ab);
        if (dot3(ab, ao) > 0.0) { memcpy(g->p[1], a, 24); g->n = 2; tri_prod(ab, ao, g->d); }
        else { memcpy(g->p[0], a, 24); g->n = 1; g->d[0] = ao[0]; g->d[1] = ao[1]; g->d[2] = ao[2]; }
    } else if (g->n == 2) {
        double c[3], b[3];
        memcpy(c, g->p[0], 24); memcpy(b, g->p[1], 24);
        gjkb_triangle(g, c, b, a);
    } else {
        double dd[3], c[3], b[3], ab[3], ac[3], ad[3], ao[3] = {-a[0], -a[1], -a[2]}, abc[3], acd[3], adb[3];
        memcpy(dd, g->p[0], 24); memcpy(c, g->p[1], 24); memcpy(b, g->p[2], 24);
        sub3(b, a, ab); sub3(c, a, ac); sub3(dd, a, ad);
        cross3(ab, ac, abc); cross3(ac, ad, acd); cross3(ad, ab, adb);
        /* outward normals: away from the opposite vertex */
        const double sabc = dot3(abc, ad) > 0.0 ? -1.0 : 1.0;
        const double sacd = dot3(acd, ab) > 0.0 ? -1.0 : 1.0;
        const double sadb = dot3(adb, ac) > 0.0 ? -1.0 : 1.0;
        if (sabc * dot3(abc, ao) > 0.0) gjkb_triangle(g, c, b, a);
        else if (sacd * dot3(acd, ao) > 0.0) gjkb_triangle(g, dd, c, a);
        else if (sadb * dot3(adb, ao) > 0.0) gjkb_triangle(g, b, dd, a);
        else return 2;
    }
    if (dot3(g->d, g->d) == 0.0) return infl ? 3 : 2;      /* the origin lies on the simplex */
    return 0;
}
static __thread long long g_gjkb_hist[40];       /* iterations of the boolean walk, per verdict (diagnostic) */
static int gjk_intersect(const core_t *A, const core_t *Bc) {
    gjkb_t g;
    gjkb_init(&g, A, Bc);
    for (;;) { const int r = gjkb_step(&g, A, Bc, 0.0); if (r) { g_gjkb_hist[g.it < 39 ? g.it : 39] += 1; return r == 2; } }
}
/* dist(A, B) < tc for tc > 0 by the inflated walk: 1 colliding, 0 free, -1 undecided */
static __thread long long g_infl_undecided = 0, g_infl_calls = 0;
static __thread long long g_infl_hist[40];
static int gjk_intersect_inflated(const core_t *A, const core_t *Bc, double tc) {
    gjkb_t g;
    gjkb_init(&g, A, Bc);
    g_infl_calls++;
    for (;;) {
        const int r = gjkb_step(&g, A, Bc, tc);
        if (r == 3) {
            g_infl_undecided++; g_infl_hist[39] += 1;
            return -1;
        }
        if (r) { g_infl_hist[g.it < 38 ? g.it : 38] += 1; return r == 2; }
    }
}
void orc_gjkb_hist(long long *out, int reset) {
    for (int i = 0; i < 40; ++i) { out[i] = g_gjkb_hist[i]; if (reset) g_gjkb_hist[i] = 0; }
}
/* out[80]: iterations of the distance predicate; out[80..119]: of the inflated walk (entry 39 = undecided) */
void orc_pred_hist(long long *out, int reset) {
    for (int i = 0; i < 80; ++i) { out[i] = g_pred_hist[i]; if (reset) g_pred_hist[i] = 0; }
    for (int i = 0; i < 40; ++i) { out[80 + i] = g_infl_hist[i]; if (reset) g_infl_hist[i] = 0; }
}

/* bounding radius of a core about its centre (broadphase) */
/* ---- exact penetration depth of overlapping cores with a cylinder or a hull: EPA ------------------------------------------
 * (the axis family above is exact for point / segment / box cores -- box-box = the 15 SAT axes -- but only an upper bound once
 * a cylinder or a hull core is involved: no rim / edge-edge directions.  Bullet runs EPA on the rounded shapes here,
 * btGjkEpaPenetrationDepthSolver; the depth of core (+) ball(m) pairs is the depth of the cores plus the margins.)
 * M = A (-) B contains the origin.  A polytope P inside M -- first the tetrahedron of M's support points along four tetrahedral
 * directions, faces oriented away from its centroid -- is grown: take the face with the smallest SIGNED distance d of its plane
 * from the origin along its outward unit normal n (negative while the origin is still outside P), ask M for its support point w
 * in direction n; n.w - d is the gap between P and M along n: converged when it is below 1e-10 (1 + n.w) with the origin inside,
 * else w becomes a vertex (the faces that see it go, the horizon is fanned to w).  Once the origin is inside,
 * d <= depth <= n.w; reported: depth = the smallest n.w seen, direction n.  No dependence on how GJK ended.
 * Returns 0 (the caller keeps the axis-family value) when the start tetrahedron is flat or the polytope breaks down. */
#define EPA_MAXIT 32
#define EPA_TOL 1e-8     /* relative gap between the inner polytope and the body along the nearest face's normal (Bullet's own EPA stops at 1e-4) */
#define EPA_MAXV (4 + EPA_MAXIT)
#define EPA_MAXF (4 + 2 * EPA_MAXIT + 8)
#define EPA_MAXE 64
typedef struct { double v[EPA_MAXV][3]; int nv; int f[EPA_MAXF][3]; double fn[EPA_MAXF][3]; double fd[EPA_MAXF]; int alive[EPA_MAXF]; int nf; double ref[3]; } epa_t;

static int epa_face_plane(const epa_t *e, int i, int j, int k, double *n, double *d) {
    double ab[3], ac[3], c[3];
    sub3(e->v[j], e->v[i], ab); sub3(e->v[k], e->v[i], ac);
    cross3(ab, ac, c);
    const double cc = dot3(c, c);
    if (!(cc > 1e-60)) return 0;
    const double inv = 1.0 / sqrt(cc);
    n[0] = c[0] * inv; n[1] = c[1] * inv; n[2] = c[2] * inv;
    *d = dot3(n, e->v[i]);
    return 1;
}
/* add face (i, j, k), oriented so that its normal points away from the reference point inside P; 0 = degenerate / no room */
static int epa_add_face(epa_t *e, int i, int j, int k) {
    double n[3], d, r[3];
    if (!epa_face_plane(e, i, j, k, n, &d)) return 0;
    sub3(e->v[i], e->ref, r);
    if (dot3(n, r) < 0.0) { const int t = j; j = k; k = t; d = -d; n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    int slot = -1;
    for (int q = 0; q < e->nf; ++q) if (!e->alive[q]) { slot = q; break; }
    if (slot < 0) { if (e->nf >= EPA_MAXF) return 0; slot = e->nf++; }
    e->f[slot][0] = i; e->f[slot][1] = j; e->f[slot][2] = k; e->fd[slot] = d; e->alive[slot] = 1;
    e->fn[slot][0] = n[0]; e->fn[slot][1] = n[1]; e->fn[slot][2] = n[2];       /* (kept: the plane is not derived again) */
    return 1;
}
static long long g_epa_calls = 0, g_epa_fail = 0, g_epa_iters = 0;
/* returns 0: no answer; 1: *depth / normal valid.  decide != 0 (the predicate at a negative contact threshold asks "deeper than x?",
 * x > 0): the loop also stops as soon as the answer is certain -- 2: the nearest face of the inner polytope is farther than x from the
 * origin (origin inside, depth >= d > x); 3: a support plane at most x away has been seen (depth <= n.w <= x) -- and kernels and
 * oracle stop at the same iteration, so the verdict is this function's by definition */
static int epa_run(const core_t *A, const core_t *Bc, double *depth, double *normal, int decide, double x) {
    __atomic_add_fetch(&g_epa_calls, 1, __ATOMIC_RELAXED);
    epa_t e;
    {   /* start tetrahedron: two support points along a fixed skew direction and its opposite, the support point farthest from
         * their line among the two along +-(a coordinate axis made perpendicular to it), the one farthest from that triangle's
         * plane among the two along +-(its normal).  Flat only when M is (both cores small against the rounding of their positions) */
        static const double D0[3] = {0.5345224838248488, -0.2672612419124244, 0.8017837257372732};
        const double nD0[3] = {-D0[0], -D0[1], -D0[2]};
        mink_support(A, Bc, D0, e.v[0]);
        mink_support(A, Bc, nD0, e.v[1]);
        double e1[3];
        sub3(e.v[1], e.v[0], e1);
        const double l1 = dot3(e1, e1);
        if (!(l1 > 1e-30)) { __atomic_add_fetch(&g_epa_fail, 1, __ATOMIC_RELAXED); return 0; }
        const double ax = fabs(e1[0]), ay = fabs(e1[1]), az = fabs(e1[2]);
        double a[3] = {0.0, 0.0, 0.0};
        if (ax <= ay && ax <= az) a[0] = 1.0; else if (ay <= az) a[1] = 1.0; else a[2] = 1.0;
        double n2[3], nn2[3], pa2[3], pb2[3], ca[3], cb[3], ra[3], rb[3];
        axpy3(-dot3(a, e1) / l1, e1, a, n2);
        nn2[0] = -n2[0]; nn2[1] = -n2[1]; nn2[2] = -n2[2];
        mink_support(A, Bc, n2, pa2);
        mink_support(A, Bc, nn2, pb2);
        sub3(pa2, e.v[0], ra); sub3(pb2, e.v[0], rb);
        cross3(e1, ra, ca); cross3(e1, rb, cb);
        const int use_a = dot3(ca, ca) >= dot3(cb, cb);
        memcpy(e.v[2], use_a ? pa2 : pb2, 24);
        double n3[3], nn3[3], pa3[3], pb3[3];
        memcpy(n3, use_a ? ca : cb, 24);
        const double l3 = dot3(n3, n3);
        if (!(l3 > 1e-24 * l1 * l1)) { __atomic_add_fetch(&g_epa_fail, 1, __ATOMIC_RELAXED); return 0; }
        nn3[0] = -n3[0]; nn3[1] = -n3[1]; nn3[2] = -n3[2];
        mink_support(A, Bc, n3, pa3);
        mink_support(A, Bc, nn3, pb3);
        sub3(pa3, e.v[0], ra); sub3(pb3, e.v[0], rb);
        const double ha = fabs(dot3(n3, ra)), hb = fabs(dot3(n3, rb));
        memcpy(e.v[3], ha >= hb ? pa3 : pb3, 24);
        const double hh = ha >= hb ? ha : hb;
        if (!(hh * hh > 1e-24 * l3 * l1)) { __atomic_add_fetch(&g_epa_fail, 1, __ATOMIC_RELAXED); return 0; }
    }
    for (int c = 0; c < 3; ++c) e.ref[c] = 0.25 * (((e.v[0][c] + e.v[1][c]) + e.v[2][c]) + e.v[3][c]);
    e.nv = 4; e.nf = 0;
    if (!epa_add_face(&e, 0, 1, 2) || !epa_add_face(&e, 0, 1, 3) || !epa_add_face(&e, 0, 2, 3) || !epa_add_face(&e, 1, 2, 3)) {
        __atomic_add_fetch(&g_epa_fail, 1, __ATOMIC_RELAXED); return 0;
    }
    double best_up = INFINITY, best_n[3] = {1.0, 0.0, 0.0};
    int ok = 0;
    for (int it = 0; it < EPA_MAXIT; ++it) {
        int bf = -1;
        for (int q = 0; q < e.nf; ++q) if (e.alive[q] && (bf < 0 || e.fd[q] < e.fd[bf])) bf = q;
        if (bf < 0) break;
        const double n[3] = {e.fn[bf][0], e.fn[bf][1], e.fn[bf][2]};
        const double d = e.fd[bf];
        if (decide && d > x) return 2;
        double w[3];
        mink_support(A, Bc, n, w);
        const double dw = dot3(n, w);
        if (dw < best_up) { best_up = dw; best_n[0] = n[0]; best_n[1] = n[1]; best_n[2] = n[2]; }
        __atomic_add_fetch(&g_epa_iters, 1, __ATOMIC_RELAXED);
        if (decide && best_up <= x) return 3;
        if (dw - d <= EPA_TOL * (1.0 + fabs(dw))) { ok = 1; break; }       /* (d < 0 here: the origin is outside M by -d: a contact within rounding) */
        if (e.nv >= EPA_MAXV) break;
        const int wi = e.nv++;
        memcpy(e.v[wi], w, 24);
        /* faces that see w go; their edges that are not shared with another such face are the horizon */
        int edges[EPA_MAXE][2], ne = 0, overflow = 0;
        for (int q = 0; q < e.nf; ++q) {
            if (!e.alive[q]) continue;
            if (dot3(e.fn[q], w) - e.fd[q] <= 0.0) continue;
            e.alive[q] = 0;
            for (int s3 = 0; s3 < 3; ++s3) {
                const int a = e.f[q][s3], b = e.f[q][(s3 + 1) % 3];
                int found = -1;
                for (int t = 0; t < ne; ++t) if ((edges[t][0] == b && edges[t][1] == a) || (edges[t][0] == a && edges[t][1] == b)) { found = t; break; }
                if (found >= 0) { edges[found][0] = edges[ne - 1][0]; edges[found][1] = edges[ne - 1][1]; --ne; }
                else if (ne < EPA_MAXE) { edges[ne][0] = a; edges[ne][1] = b; ++ne; }
                else overflow = 1;
            }
        }
        if (ne < 3 || overflow) break;
        int bad = 0;
        for (int t = 0; t < ne; ++t) if (!epa_add_face(&e, edges[t][0], edges[t][1], wi)) { bad = 1; break; }
        if (bad) break;
    }
    if (!(best_up < INFINITY)) { __atomic_add_fetch(&g_epa_fail, 1, __ATOMIC_RELAXED); return 0; }
    if (!ok) __atomic_add_fetch(&g_epa_fail, 1, __ATOMIC_RELAXED);
    *depth = best_up > 0.0 ? best_up : 0.0;
    normal[0] = -best_n[0]; normal[1] = -best_n[1]; normal[2] = -best_n[2];      /* from B to A */
    return 1;
}
static int epa_depth(const core_t *A, const core_t *Bc, double *depth, double *normal) { return epa_run(A, Bc, depth, normal, 0, 0.0); }
void orc_epa_stats(long long *out, int reset) {
    out[0] = g_epa_calls; out[1] = g_epa_fail; out[2] = g_epa_iters;
    if (reset) { g_epa_calls = 0; g_epa_fail = 0; g_epa_iters = 0; }
}
/* depth and direction (from B to A) of two overlapping cores: the axis family, tightened by EPA where the family is only a bound */
static double overlap_depth_exact(const core_t *A, const core_t *Bc, double *normal) {
    double depth = overlap_depth(A, Bc, normal);
    if (A->kind == K_CYL || A->kind == K_HULL || Bc->kind == K_CYL || Bc->kind == K_HULL) {
        double de, ne[3];
        if (epa_depth(A, Bc, &de, ne) && de < depth) { depth = de; normal[0] = ne[0]; normal[1] = ne[1]; normal[2] = ne[2]; }
    }
    return depth;
}

/* is the exact depth of two overlapping cores larger than x?  = (overlap_depth_exact > x), without running EPA to the end:
 * the family's value is an upper bound (not deeper than x: done), exact without a cylinder / hull core, and EPA stops at its
 * first certain answer */
static int overlap_deeper_than(const core_t *A, const core_t *Bc, double x) {
    double nrm[3];
    /* a hull's family scans faces x vertices: there EPA goes first and the family is consulted only when EPA leaves the question open;
     * everywhere else the family is a handful of axes and settles most items before a polytope is built */
    const int hull = A->kind == K_HULL || Bc->kind == K_HULL;
    if (!hull) {
        const double fam = overlap_depth(A, Bc, nrm);
        if (!(fam > x)) return 0;
        if (!(A->kind == K_CYL || Bc->kind == K_CYL)) return 1;
    }
    double de = 0.0, ne[3];
    const int r = epa_run(A, Bc, &de, ne, 1, x);
    if (r == 2) return 1;
    if (r == 3) return 0;
    if (hull) {
        const double fam = overlap_depth(A, Bc, nrm);
        if (!(fam > x)) return 0;
    }
    if (r == 1) return de > x;
    return 1;                                  /* no answer from EPA: the family's value stands */
}

static double core_bound_radius(const core_t *s) {
    switch (s->kind) {
        case K_POINT: return 0.0;
        case K_SEG: return s->h[0];
        case K_CYL: return sqrt(FMA(s->rad, s->rad, s->h[0] * s->h[0]));
        case K_BOX: return sqrt(FMA(s->h[2], s->h[2], FMA(s->h[1], s->h[1], s->h[0] * s->h[0])));
        case K_HULL: return s->rho;
        default: return INFINITY;
    }
}

static __thread long long g_stat_items = 0, g_stat_survive = 0, g_stat_gjk = 0;   /* per thread: no sharing */

/* predicate used by validity: is the signed distance below thr?
 *   0. planes (always the second shape): t = thr + mA, height hc = n.(cA - p0); hc - rhoA >= t => free
 *      (broadphase), else hc - halfwidth_A(n) < t;
 *   1. decided on the CORE distance against tc = (thr + mA) + mB;
 *   2. broadphase: bounding spheres -- |cA - cB|^2 >= ((tc + rhoA) + rhoB)^2 (or a non-positive sum) => free;
 *   3. the two cores are taken in canonical order (kind ascending: point < segment < box < cylinder), so that
 *      a device can evaluate all pairs of one kind class with one specialised routine;
 *   4. midphase when a box core is involved, on the OTHER core's centre c (a point of that core), without
 *      square roots: d2 = squared distance of c to the box; outside: d2 >= (tc + rho_other)^2 (tc >= 0) => free,
 *      d2 < tc^2 (tc > 0) => colliding; inside at depth g: -g < tc => colliding;
 *   5. exact test: closed form for point/segment cores and point-vs-solid; otherwise GJK -- the boolean walk
 *      (gjk_intersect) when tc == 0; for tc > 0 the same walk with core A inflated by a ball of radius tc
 *      (gjk_intersect_inflated: "A (+) ball(tc) meets B" is "dist < tc"), and the distance iteration with early exits
 *      (gjk_collides) for tc < 0, for every pair with a hull core, and for the few pairs the inflated walk leaves
 *      undecided after 64 steps. */
static int cores_collide(const core_t *A0, const core_t *B0, double thr) {
    if (B0->kind == K_PLANE) {
        double d[3];
        sub3(A0->c, B0->c, d);
        const double hc = dot3(d, B0->ax[2]);
        const double t = thr + A0->margin;
        if ((hc - core_bound_radius(A0)) >= t) return 0;       /* broadphase: bounding sphere above the plane */
        double hw, hpos;
        core_extents(A0, B0->ax[2], &hw, &hpos);
        return (hc - hw) < t;
    }
    const double tc = (thr + A0->margin) + B0->margin;
    const double rs = (tc + core_bound_radius(A0)) + core_bound_radius(B0);
    double dl[3];
    sub3(A0->c, B0->c, dl);
    g_stat_items++;
    if (!(rs > 0.0)) return 0;
    if (dot3(dl, dl) >= rs * rs) return 0;
    g_stat_survive++;
    const core_t *A = A0, *Bc = B0;
    if (A0->kind > B0->kind) { A = B0; Bc = A0; }
    /* midphase for box cores: the other core's CENTRE (a point of that core) against the exact box.
     * centre farther than the bounding radius => free; centre closer than tc => colliding. */
    {
        const core_t *bx = NULL, *ot = NULL;
        if (Bc->kind == K_BOX) { bx = Bc; ot = A; }
        else if (A->kind == K_BOX) { bx = A; ot = Bc; }
        if (bx) {
            /* square-root free: x = centre in box coordinates, ex = excess beyond the faces */
            double d[3], ax[3], ex[3];
            int inside = 1;
            sub3(ot->c, bx->c, d);
            for (int j = 0; j < 3; ++j) {
                ax[j] = fabs(dot3(d, bx->ax[j]));
                ex[j] = ax[j] - bx->h[j];
                if (ex[j] > 0.0) inside = 0; else ex[j] = 0.0;
            }
            const double d2 = FMA(ex[2], ex[2], FMA(ex[1], ex[1], ex[0] * ex[0]));
            if (!inside) {
                if (tc >= 0.0) { const double r = tc + core_bound_radius(ot); if (d2 >= r * r) return 0; }
                if (tc > 0.0 && d2 < tc * tc) return 1;
            } else {
                double g = bx->h[0] - ax[0];
                if (bx->h[1] - ax[1] < g) g = bx->h[1] - ax[1];
                if (bx->h[2] - ax[2] < g) g = bx->h[2] - ax[2];
                if (-g < tc) return 1;
            }
        }
    }
    const int a_ps = (A->kind == K_POINT || A->kind == K_SEG), b_ps = (Bc->kind == K_POINT || Bc->kind == K_SEG);
    if (a_ps && b_ps) {
        double pa[3], pb[3], e[3];
        ps_closest(A, Bc, pa, pb);
        sub3(pa, pb, e);
        return sqrt(dot3(e, e)) < tc;
    }
    if (A->kind == K_POINT && Bc->kind != K_HULL) { double cp[3], nb[3]; return point_solid(A->c, Bc, cp, nb) < tc; }
    g_stat_gjk++;
    if (A->kind != K_HULL && Bc->kind != K_HULL) {
        /* (not for hulls: every step scans a vertex list, and the distance iteration's early exits need half as many steps) */
        if (tc == 0.0) return gjk_intersect(A, Bc);  /* pure intersection test: the boolean walk */
        if (tc > 0.0) {                              /* the same walk with core A inflated by tc ... */
            const int r = gjk_intersect_inflated(A, Bc, tc);
            if (r >= 0) return r;
        }
    }
    return gjk_collides(A, Bc, tc);                  /* ... and the distance iteration for tc < 0 or an undecided walk */
}

/* diagnostic: one pair of one configuration, with the walk of either predicate printed (tools/fuzz_repro.py) */
static void robot_cores(const orc_model *m, const double *q, xf_t *frames, core_t *rc);
static core_t *build_world_cores(const orc_model *m);
int orc_pair_trace(const orc_model *m, const double *q, int32_t p, double thr) {
    core_t *wc = build_world_cores(m);
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(m->n_joints > 0 ? m->n_joints : 1));
    core_t *rc = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_rshapes > 0 ? m->n_rshapes : 1));
    robot_cores(m, q, frames, rc);
    const int a = m->pair_a[p], b = m->pair_b[p];
    const core_t *A = &rc[a], *Bc = b < m->n_rshapes ? &rc[b] : &wc[b - m->n_rshapes];
    if (A->kind > Bc->kind) { const core_t *t = A; A = Bc; Bc = t; }
    const double tc = (thr + A->margin) + Bc->margin;
    fprintf(stderr, "pair %d: kinds %d %d  tc %.17g  margins %.6g %.6g  rad %.6g %.6g  h %.6g/%.6g/%.6g %.6g/%.6g/%.6g  hull verts %d %d\n", p, A->kind, Bc->kind, tc,
            A->margin, Bc->margin, A->rad, Bc->rad, A->h[0], A->h[1], A->h[2], Bc->h[0], Bc->h[1], Bc->h[2], A->hn, Bc->hn);
    for (int side = 0; side < 2; ++side) {
        const core_t *s = side ? Bc : A;
        fprintf(stderr, "  core%d c %.17g %.17g %.17g\n", side, s->c[0], s->c[1], s->c[2]);
        for (int j = 0; j < 3; ++j) fprintf(stderr, "  core%d ax%d %.17g %.17g %.17g\n", side, j, s->ax[j][0], s->ax[j][1], s->ax[j][2]);
        for (int k = 0; k < s->hn; ++k) fprintf(stderr, "  core%d v %.17g %.17g %.17g\n", side, s->hv[3 * k], s->hv[3 * k + 1], s->hv[3 * k + 2]);
        for (int k = 0; k < s->hf; ++k) fprintf(stderr, "  core%d f %.17g %.17g %.17g %.17g\n", side, s->hp[4 * k], s->hp[4 * k + 1], s->hp[4 * k + 2], s->hp[4 * k + 3]);
    }
    double wit[9];
    int it = 0;
    const double dcan = cores_distance(A, Bc, wit, &it);
    fprintf(stderr, "  distance (canonical order) %.17g after %d iterations\n", dcan, it);
    fprintf(stderr, "  distance (swapped)         %.17g\n", cores_distance(Bc, A, wit, &it));
    gjkb_t g;
    gjkb_init(&g, A, Bc);
    for (;;) {
        double w[3];
        mink_support(A, Bc, g.d, w);
        fprintf(stderr, "  bool it %2d n %d d (%.6g %.6g %.6g) w (%.9g %.9g %.9g) w.d %.6g\n", g.it, g.n, g.d[0], g.d[1], g.d[2], w[0], w[1], w[2], dot3(w, g.d));
        { double sa[3], sb[3], nd[3] = {-g.d[0], -g.d[1], -g.d[2]}; core_support(A, g.d, sa); core_support(Bc, nd, sb);
          double r[3]; sub3(sa, A->c, r);
          fprintf(stderr, "      d %.17g %.17g %.17g | sa-c axial %.17g |sa-c| %.17g  d.u %.17g\n", g.d[0], g.d[1], g.d[2], dot3(r, A->ax[2]), sqrt(dot3(r, r)), dot3(g.d, A->ax[2])); }
        for (int i = 0; i < g.n; ++i) fprintf(stderr, "      p%d (%.9g %.9g %.9g)\n", i, g.p[i][0], g.p[i][1], g.p[i][2]);
        const int r = gjkb_step(&g, A, Bc, 0.0);
        if (r) { fprintf(stderr, "  boolean walk: verdict %d after %d iterations\n", r, g.it); break; }
    }
    fprintf(stderr, "  gjk_collides(tc) = %d, cores_collide = %d\n", gjk_collides(A, Bc, tc), cores_collide(A, Bc, thr));
    free(wc); free(frames); free(rc);
    return 0;
}

void orc_stats(long long *out, int reset) {
    out[0] = g_stat_items; out[1] = g_stat_survive; out[2] = g_stat_gjk; out[3] = g_infl_calls; out[4] = g_infl_undecided;
    if (reset) { g_stat_items = g_stat_survive = g_stat_gjk = 0; g_infl_calls = g_infl_undecided = 0; }
}

double orc_shape_distance_m(const orc_model *hulls, int32_t type_a, const double *pose_a, const double *param_a, int32_t type_b,
                            const double *pose_b, const double *param_b, double *witness, int32_t *iters) {
    xf_t xa, xb;
    core_t A, Bc;
    xf_from12(pose_a, &xa); xf_from12(pose_b, &xb);
    core_from_shape(hulls, type_a, &xa, param_a, &A);
    core_from_shape(hulls, type_b, &xb, param_b, &Bc);
    int it = 0;
    const double d = cores_distance(&A, &Bc, witness, &it);
    if (iters) *iters = it;
    return d;
}

double orc_shape_distance(int32_t type_a, const double *pose_a, const double *param_a, int32_t type_b,
                          const double *pose_b, const double *param_b, double *witness, int32_t *iters) {
    return orc_shape_distance_m(NULL, type_a, pose_a, param_a, type_b, pose_b, param_b, witness, iters);
}

/* ------------------------------------------------------------------------------------------------
 * per-configuration collision evaluation
 * ---------------------------------------------------------------------------------------------- */
typedef struct { core_t *world; } scene_cache_t;

static core_t *build_world_cores(const orc_model *m) {
    core_t *w = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_wshapes > 0 ? m->n_wshapes : 1));
    for (int i = 0; i < m->n_wshapes; ++i) {
        xf_t x;
        xf_from12(m->wshape_pose + 12 * i, &x);
        core_from_shape(m, m->wshape_type[i], &x, m->wshape_param + 4 * i, &w[i]);
    }
    return w;
}

/* frames of all joints (topological order), then world cores of all robot shapes */
static void robot_cores(const orc_model *m, const double *q, xf_t *frames, core_t *rc) {
    xf_t base;
    xf_from12(m->base_pose, &base);
    for (int k = 0; k < m->n_joints; ++k) {
        const xf_t *par = m->joint_parent[k] < 0 ? &base : &frames[m->joint_parent[k]];
        joint_apply(m, k, par, q, &frames[k]);
    }
    for (int s = 0; s < m->n_rshapes; ++s) {
        const xf_t *F = m->rshape_frame[s] < 0 ? &base : &frames[m->rshape_frame[s]];
        xf_t loc, W;
        xf_from12(m->rshape_local + 12 * s, &loc);
        xf_mul(F, loc.R, loc.t, &W);
        core_from_shape(m, m->rshape_type[s], &W, m->rshape_param + 4 * s, &rc[s]);
    }
}

static int pair_hit(const orc_model *m, const core_t *rc, const core_t *wc, int p, double thr) {
    const int a = m->pair_a[p], b = m->pair_b[p];
    return cores_collide(&rc[a], b < m->n_rshapes ? &rc[b] : &wc[b - m->n_rshapes], thr);
}

static double pair_eval(const orc_model *m, const core_t *rc, const core_t *wc, int p, double *wit) {
    const int a = m->pair_a[p], b = m->pair_b[p];
    const core_t *A = &rc[a];
    const core_t *Bc = b < m->n_rshapes ? &rc[b] : &wc[b - m->n_rshapes];
    return cores_distance(A, Bc, wit, NULL);
}

int orc_pair_distances(const orc_model *m, const double *q, int64_t B, double *dist, double *witness) {
    core_t *wc = build_world_cores(m);
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(m->n_joints > 0 ? m->n_joints : 1));
    core_t *rc = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_rshapes > 0 ? m->n_rshapes : 1));
    for (int64_t b = 0; b < B; ++b) {
        robot_cores(m, q + b * m->n_q, frames, rc);
        for (int p = 0; p < m->n_pairs; ++p)
            dist[b * m->n_pairs + p] = pair_eval(m, rc, wc, p, witness ? witness + (b * m->n_pairs + p) * 9 : NULL);
    }
    free(wc); free(frames); free(rc);
    return 0;
}

/* bit k set = joint k lies on the path from the base to moving frame f */
static unsigned frame_mask(const orc_model *m, int f) {
    unsigned mk = 0u;
    while (f >= 0) { mk |= 1u << f; f = m->joint_parent[f]; }
    return mk;
}

int orc_proximity_jacobian(const orc_model *m, const double *q, int64_t B, double *dist, double *witness, double *jrows) {
    core_t *wc = build_world_cores(m);
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(m->n_joints > 0 ? m->n_joints : 1));
    core_t *rc = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_rshapes > 0 ? m->n_rshapes : 1));
    const int nq = m->n_q;
    for (int64_t b = 0; b < B; ++b) {
        robot_cores(m, q + b * nq, frames, rc);
        for (int p = 0; p < m->n_pairs; ++p) {
            double *wit = witness + (b * m->n_pairs + p) * 9;
            double *row = jrows + (b * m->n_pairs + p) * nq;
            dist[b * m->n_pairs + p] = pair_eval(m, rc, wc, p, wit);
            for (int c = 0; c < nq; ++c) row[c] = 0.0;
            const int sa = m->pair_a[p], sb = m->pair_b[p];
            const unsigned ma = frame_mask(m, m->rshape_frame[sa]);
            const unsigned mb = sb < m->n_rshapes ? frame_mask(m, m->rshape_frame[sb]) : 0u;
            for (int k = 0; k < m->n_joints; ++k) {
                const int in_a = (ma >> k) & 1u, in_b = (mb >> k) & 1u;
                if (!in_a && !in_b) continue;
                const double *a = m->joint_axis + 3 * k;
                const xf_t *F = &frames[k];
                double w[3];
                for (int r = 0; r < 3; ++r) w[r] = FMA(F->R[3 * r + 2], a[2], FMA(F->R[3 * r + 1], a[1], F->R[3 * r] * a[0]));
                const int rev = m->joint_type[k] == ORC_REVOLUTE;
                double va = 0.0, vb = 0.0;
                if (in_a) {
                    if (rev) { double dd[3], v[3]; sub3(wit, F->t, dd); cross3(w, dd, v); va = dot3(wit + 6, v); }
                    else va = dot3(wit + 6, w);
                }
                if (in_b) {
                    if (rev) { double dd[3], v[3]; sub3(wit + 3, F->t, dd); cross3(w, dd, v); vb = dot3(wit + 6, v); }
                    else vb = dot3(wit + 6, w);
                }
                row[m->joint_qidx[k]] = va - vb;
            }
        }
    }
    free(wc); free(frames); free(rc);
    return 0;
}

int orc_closest(const orc_model *m, const double *q, int64_t B, double *min_dist, int32_t *argmin) {
    core_t *wc = build_world_cores(m);
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(m->n_joints > 0 ? m->n_joints : 1));
    core_t *rc = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_rshapes > 0 ? m->n_rshapes : 1));
    for (int64_t b = 0; b < B; ++b) {
        robot_cores(m, q + b * m->n_q, frames, rc);
        double best = INFINITY;
        int bi = -1;
        for (int p = 0; p < m->n_pairs; ++p) {
            const double d = pair_eval(m, rc, wc, p, NULL);
            if (d < best) { best = d; bi = p; }
        }
        min_dist[b] = best;
        if (argmin) argmin[b] = bi;
    }
    free(wc); free(frames); free(rc);
    return 0;
}

typedef struct {
    const orc_model *m; const double *q; int64_t b0, b1; double thr; uint8_t *mask; const core_t *wc;
} vjob_t;

/* a configuration with a NaN or infinite joint value counts as colliding (a planner must not accept it) */
static int q_nonfinite(const double *q, int n) {
    for (int j = 0; j < n; ++j) if (!(fabs(q[j]) <= DBL_MAX)) return 1;
    return 0;
}

static void *validity_worker(void *arg) {
    vjob_t *j = (vjob_t *)arg;
    const orc_model *m = j->m;
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(m->n_joints > 0 ? m->n_joints : 1));
    core_t *rc = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_rshapes > 0 ? m->n_rshapes : 1));
    for (int64_t b = j->b0; b < j->b1; ++b) {
        if (q_nonfinite(j->q + b * m->n_q, m->n_q)) { j->mask[b] = 1; continue; }
        robot_cores(m, j->q + b * m->n_q, frames, rc);
        uint8_t hit = 0;
        for (int p = 0; p < m->n_pairs && !hit; ++p)
            if (pair_hit(m, rc, j->wc, p, j->thr)) hit = 1;
        j->mask[b] = hit;
    }
    free(frames); free(rc);
    return NULL;
}

int orc_validity(const orc_model *m, const double *q, int64_t B, double threshold, uint8_t *mask, int32_t nthreads) {
    core_t *wc = build_world_cores(m);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    vjob_t jobs[256];
    const int64_t chunk = (B + nthreads - 1) / nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t b0 = t * chunk, b1 = (b0 + chunk < B) ? b0 + chunk : B;
        if (b0 >= B) break;
        jobs[t] = (vjob_t){m, q, b0, b1, threshold, mask, wc};
        if (nthreads == 1) validity_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, validity_worker, &jobs[t]);
        ++started;
    }
    if (nthreads > 1) for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(wc);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * DiscreteConnector (connectors.py:57-100).  T = arange(0, T_f, res/d) then T_f appended;
 * arange length = ceil(T_f / step), values i*step; traj(t) = (1-t)*start + t*goal (unfused).
 * ---------------------------------------------------------------------------------------------- */
static double edge_length(int32_t n_q, const double *s, const double *g) {
    double acc = 0.0;
    for (int i = 0; i < n_q; ++i) { const double df = g[i] - s[i]; acc = FMA(df, df, acc); }
    return sqrt(acc);
}
static void edge_point(int32_t n_q, const double *s, const double *g, double t, double *o) {
    const double omt = 1.0 - t;
    for (int i = 0; i < n_q; ++i) { const double a = omt * s[i]; const double b = t * g[i]; o[i] = a + b; }
}
/* returns n = len(arange) (so n+1 samples) or -1 for the degenerate edge; writes T_f and step */
static int64_t edge_plan(double d, double resolution, double max_distance, int mode, double *Tf, double *step) {
    if (!(d > (double)FLT_EPSILON && d <= DBL_MAX)) return -1;      /* too short (the reference returns None) or not finite */
    *Tf = (mode == 1 && d > max_distance) ? max_distance / d : 1.0;
    *step = resolution / d;
    const double len = ceil(*Tf / *step);
    return len > 0.0 ? (int64_t)len : 0;
}

int orc_edge_samples(int32_t n_q, const double *start, const double *goal, double dist, double resolution,
                     double max_distance, int32_t mode, double *out, int32_t max_samples) {
    const double d = dist >= 0.0 ? dist : edge_length(n_q, start, goal);
    double Tf, step;
    const int64_t n = edge_plan(d, resolution, max_distance, mode, &Tf, &step);
    if (n < 0) return 0;
    int cnt = 0;
    for (int64_t i = 0; i <= n && cnt < max_samples; ++i, ++cnt)
        edge_point(n_q, start, goal, i < n ? (double)i * step : Tf, out + (size_t)cnt * n_q);
    return (int)(n + 1);
}

typedef struct {
    const orc_model *m; const double *starts, *goals, *dist; int64_t e0, e1; double res, maxd, thr; int mode;
    uint8_t *valid; double *end; int32_t *ns; const core_t *wc;
} ejob_t;

static void *edge_worker(void *arg) {
    ejob_t *j = (ejob_t *)arg;
    const orc_model *m = j->m;
    const int nq = m->n_q;
    xf_t *frames = (xf_t *)malloc(sizeof(xf_t) * (size_t)(m->n_joints > 0 ? m->n_joints : 1));
    core_t *rc = (core_t *)malloc(sizeof(core_t) * (size_t)(m->n_rshapes > 0 ? m->n_rshapes : 1));
    double *qs = (double *)malloc(sizeof(double) * (size_t)nq);
    for (int64_t e = j->e0; e < j->e1; ++e) {
        const double *s = j->starts + e * nq, *g = j->goals + e * nq;
        const double d = j->dist ? j->dist[e] : edge_length(nq, s, g);
        double Tf, step;
        const int64_t n = edge_plan(d, j->res, j->maxd, j->mode, &Tf, &step);
        if (n < 0) {
            j->valid[e] = 0;
            if (j->ns) j->ns[e] = 0;
            if (j->end) for (int i = 0; i < nq; ++i) j->end[e * nq + i] = NAN;
            continue;
        }
        uint8_t ok = 1;
        for (int64_t i = 0; i <= n && ok; ++i) {
            edge_point(nq, s, g, i < n ? (double)i * step : Tf, qs);
            if (q_nonfinite(qs, nq)) { ok = 0; break; }
            robot_cores(m, qs, frames, rc);
            for (int p = 0; p < m->n_pairs; ++p)
                if (pair_hit(m, rc, j->wc, p, j->thr)) { ok = 0; break; }
        }
        j->valid[e] = ok;
        if (j->ns) j->ns[e] = (int32_t)(n + 1);
        if (j->end) {
            if (j->mode == 0) for (int i = 0; i < nq; ++i) j->end[e * nq + i] = g[i];
            else edge_point(nq, s, g, Tf, j->end + e * nq);
        }
    }
    free(frames); free(rc); free(qs);
    return NULL;
}

int orc_edge_validity(const orc_model *m, const double *starts, const double *goals, const double *dist,
                      int64_t E, double resolution, double max_distance, int32_t mode, double threshold,
                      uint8_t *valid, double *end, int32_t *n_samples, int32_t nthreads) {
    core_t *wc = build_world_cores(m);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    ejob_t jobs[256];
    const int64_t chunk = (E + nthreads - 1) / nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t e0 = t * chunk, e1 = (e0 + chunk < E) ? e0 + chunk : E;
        if (e0 >= E) break;
        jobs[t] = (ejob_t){m, starts, goals, dist, e0, e1, resolution, max_distance, threshold, mode, valid, end, n_samples, wc};
        if (nthreads == 1) edge_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, edge_worker, &jobs[t]);
        ++started;
    }
    if (nthreads > 1) for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(wc);
    return 0;
}

int orc_shape_collides_m(const orc_model *hulls, int32_t type_a, const double *pose_a, const double *param_a, int32_t type_b,
                         const double *pose_b, const double *param_b, double threshold) {
    xf_t xa, xb;
    core_t A, Bc;
    xf_from12(pose_a, &xa); xf_from12(pose_b, &xb);
    core_from_shape(hulls, type_a, &xa, param_a, &A);
    core_from_shape(hulls, type_b, &xb, param_b, &Bc);
    return cores_collide(&A, &Bc, threshold);
}

int orc_shape_collides(int32_t type_a, const double *pose_a, const double *param_a, int32_t type_b,
                       const double *pose_b, const double *param_b, double threshold) {
    return orc_shape_collides_m(NULL, type_a, pose_a, param_a, type_b, pose_b, param_b, threshold);
}
