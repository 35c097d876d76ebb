// nbk_device.hpp -- per-lane device math for the gfx950 kernels: transforms, sincos, convex cores,
// support mapping, GJK, overlap depth, closed-form point/segment distances.
//
// One configuration per lane; everything here is straight-line float64 on VGPRs.  Shape kinds and all
// model constants are wave-uniform, so every `switch (kind)` below is a scalar branch (no divergence);
// the only divergent control flow is the GJK iteration count and its Voronoi-region walk.
//
// Arithmetic contract: compiled with -ffp-contract=off, every fused multiply-add is written
// as NBK_FMA, sqrt/divide are the IEEE correctly-rounded ones, and sin/cos are nbk_sincos below -- so the
// results are a pure function of the written operation order and can be checked bit-for-bit on a CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NBK_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define NBK_DEV __device__ __forceinline__
#define NBK_INF __builtin_huge_val()

namespace nbk {

// K_HULL: convex hull of a vertex list in the primitive's local frame (MESH shapes: one hull per mesh object, which is what
// the reference's GEOM_MESH collision shapes are in Bullet; numbotics/utils/shape.py:81-94, numbotics/utils/mesh.py:18-37).
enum { K_POINT = 0, K_SEG = 1, K_BOX = 2, K_CYL = 3, K_HULL = 4, K_PLANE = 5 };

// ---- small vectors ---------------------------------------------------------------------------
NBK_DEV double dot3(const double* a, const double* b) { return NBK_FMA(a[2], b[2], NBK_FMA(a[1], b[1], a[0] * b[0])); }
NBK_DEV void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
NBK_DEV void cross3(const double* a, const double* b, double* o) {
    o[0] = NBK_FMA(a[1], b[2], -(a[2] * b[1]));
    o[1] = NBK_FMA(a[2], b[0], -(a[0] * b[2]));
    o[2] = NBK_FMA(a[0], b[1], -(a[1] * b[0]));
}
// o = a + s*b   (o may alias a)
NBK_DEV void axpy3(double s, const double* b, const double* a, double* o) {
    const double o0 = NBK_FMA(s, b[0], a[0]), o1 = NBK_FMA(s, b[1], a[1]), o2 = NBK_FMA(s, b[2], a[2]);
    o[0] = o0; o[1] = o1; o[2] = o2;
}
NBK_DEV void copy3(const double* a, double* o) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
NBK_DEV double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
NBK_DEV double nbk_sqrt(double x) { return __builtin_sqrt(x); }

// ---- 3x4 rigid transforms ----------------------------------------------------------------------
struct Xf { double R[9]; double t[3]; };

// o = a * (Rb, tb); Rb/tb may be wave-uniform constants
NBK_DEV void xf_mul(const Xf& a, const double* Rb, const double* tb, Xf& o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a0 = a.R[3 * i], a1 = a.R[3 * i + 1], a2 = a.R[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) o.R[3 * i + j] = NBK_FMA(a2, Rb[6 + j], NBK_FMA(a1, Rb[3 + j], a0 * Rb[j]));
        o.t[i] = NBK_FMA(a2, tb[2], NBK_FMA(a1, tb[1], NBK_FMA(a0, tb[0], a.t[i])));
    }
}
// only column j of the rotation of a*(Rb,.)
NBK_DEV void xf_mul_col(const Xf& a, const double* Rb, int j, double* col) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        col[i] = NBK_FMA(a.R[3 * i + 2], Rb[6 + j], NBK_FMA(a.R[3 * i + 1], Rb[3 + j], a.R[3 * i] * Rb[j]));
}
NBK_DEV void xf_mul_pos(const Xf& a, const double* tb, double* p) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        p[i] = NBK_FMA(a.R[3 * i + 2], tb[2], NBK_FMA(a.R[3 * i + 1], tb[1], NBK_FMA(a.R[3 * i], tb[0], a.t[i])));
}
NBK_DEV void xf_from12(const double* p, Xf& x) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x.R[3 * i] = p[4 * i]; x.R[3 * i + 1] = p[4 * i + 1]; x.R[3 * i + 2] = p[4 * i + 2];
        x.t[i] = p[4 * i + 3];
    }
}

// ---- sincos: Cody-Waite reduction by pi/2 + minimax kernels on [-pi/4, pi/4] --------------------
NBK_DEV void nbk_sincos(double x, double& s, double& c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_2 = 6.07710050630396597660e-11,
                 PIO2_3 = 2.02226624871116645580e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    if (!(__builtin_fabs(x) < 2147483648.0)) { s = __builtin_nan(""); c = s; return; }
    const double k = __builtin_rint(x * TWO_OVER_PI);
    double r = NBK_FMA(-k, PIO2_1, x);
    r = NBK_FMA(-k, PIO2_2, r);
    r = NBK_FMA(-k, PIO2_3, r);
    const double z = r * r;
    double ps = NBK_FMA(z, S6, S5);
    ps = NBK_FMA(z, ps, S4); ps = NBK_FMA(z, ps, S3); ps = NBK_FMA(z, ps, S2); ps = NBK_FMA(z, ps, S1);
    const double sr = NBK_FMA(r * z, ps, r);
    double pc = NBK_FMA(z, C6, C5);
    pc = NBK_FMA(z, pc, C4); pc = NBK_FMA(z, pc, C3); pc = NBK_FMA(z, pc, C2); pc = NBK_FMA(z, pc, C1);
    const double cr = NBK_FMA(z * z, pc, NBK_FMA(-0.5, z, 1.0));
    const int n = (int)((long long)k & 3LL);
    const bool swap = (n & 1) != 0;
    const double a = swap ? cr : sr;       // |sin|
    const double b = swap ? sr : cr;       // |cos|
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b : b;
}

// ---- convex cores ------------------------------------------------------------------------------
// shape = core (+) ball(margin):  sphere = point, capsule = segment, box, cylinder (axis = local z), hull (vertex list)
struct HullRef { const double* hv; const double* hp; int hn; int hf; };   // vertices [hn][3], face planes [hf][4] (n, d), local frame;
                                                                         // hv[-6..-1] = the hull's local bounding box: centre (3), half extents (3)
struct Core {
    int kind;            // wave-uniform
    double c[3];
    double ax[3][3];     // ax[j] = world direction of local axis j (box, hull: all three; seg/cyl: ax[2])
    double h[3];         // uniform: seg/cyl h[0] = half length; box half extents; K_HULL: the 24 bytes of a HullRef (the shape
                         // tables hold the hull's device pointers and counts there), read through hull_* below.  Not a union:
                         // with one the compiler kept h[] of both cores in scratch, and every box support read it from there
    double rad;          // uniform: cylinder radius
    double margin;       // uniform
    double rho;          // uniform: bounding radius of the core about c
};
NBK_DEV const double* hull_hv(const Core& s) { return reinterpret_cast<const double*>(__builtin_bit_cast(unsigned long long, s.h[0])); }
NBK_DEV const double* hull_hp(const Core& s) { return reinterpret_cast<const double*>(__builtin_bit_cast(unsigned long long, s.h[1])); }
NBK_DEV int hull_hn(const Core& s) { return (int)(unsigned)(__builtin_bit_cast(unsigned long long, s.h[2]) & 0xFFFFFFFFull); }
NBK_DEV int hull_hf(const Core& s) { return (int)(unsigned)(__builtin_bit_cast(unsigned long long, s.h[2]) >> 32); }

typedef const __attribute__((address_space(3))) double* LdsDoubleP;
// four vertices per trip: their twelve loads are issued together (per-lane loads in k_narrow, where the lanes of a wave hold
// different hulls), the comparisons stay in vertex order
template <typename P>
NBK_DEV void hull_first_max(P hv, int hn, double dl0, double dl1, double dl2, double& v0, double& v1, double& v2) {
    double best = -NBK_INF;
    int bi = 0;
    int k = 0;
    for (; k + 4 <= hn; k += 4) {
        double vx[4], vy[4], vz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { vx[u] = hv[3 * (k + u)]; vy[u] = hv[3 * (k + u) + 1]; vz[u] = hv[3 * (k + u) + 2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double pr = NBK_FMA(vz[u], dl2, NBK_FMA(vy[u], dl1, vx[u] * dl0));
            if (pr > best) { best = pr; bi = k + u; }
        }
    }
    for (; k < hn; ++k) {
        const double pr = NBK_FMA(hv[3 * k + 2], dl2, NBK_FMA(hv[3 * k + 1], dl1, hv[3 * k] * dl0));
        if (pr > best) { best = pr; bi = k; }
    }
    v0 = hv[3 * bi]; v1 = hv[3 * bi + 1]; v2 = hv[3 * bi + 2];
}

template <typename P>
NBK_DEV void hull_min_max(P hv, int hn, double dl0, double dl1, double dl2, double& neg, double& pos) {
    double hi = -NBK_INF, lo = NBK_INF;
    for (int k = 0; k < hn; ++k) {
        const double pr = NBK_FMA(hv[3 * k + 2], dl2, NBK_FMA(hv[3 * k + 1], dl1, hv[3 * k] * dl0));
        if (pr > hi) hi = pr;
        if (pr < lo) lo = pr;
    }
    pos = hi; neg = -lo;
}

NBK_DEV void core_support(const Core& s, const double* d, double* o) {
    switch (s.kind) {
        case K_POINT: copy3(s.c, o); break;
        case K_SEG: {
            const double du = dot3(d, s.ax[2]);
            const double sg = du >= 0.0 ? s.h[0] : -s.h[0];
            axpy3(sg, s.ax[2], s.c, o);
        } break;
        case K_CYL: {
            const double du = dot3(d, s.ax[2]);
            const double sg = du >= 0.0 ? s.h[0] : -s.h[0];
            // w = (u.u) d - (d.u) u, twice: perpendicular to the axis whatever its length (see the oracle: a joint axis given to
            // five digits leaves the link rotations orthonormal to 1e-6 only)
            const double uu = dot3(s.ax[2], s.ax[2]);
            double w[3], t[3];
            t[0] = uu * d[0]; t[1] = uu * d[1]; t[2] = uu * d[2];
            axpy3(-du, s.ax[2], t, w);
            const double wu = dot3(w, s.ax[2]);
            t[0] = uu * w[0]; t[1] = uu * w[1]; t[2] = uu * w[2];
            axpy3(-wu, s.ax[2], t, w);
            const double ww = dot3(w, w);
            axpy3(sg, s.ax[2], s.c, o);
            // (a direction axial to 1e-13 has no radial part worth the name: see the oracle)
            const double u4 = (uu * uu) * (uu * uu);
            if (ww > (1e-26 * u4) * dot3(d, d)) {
                const double k = s.rad / nbk_sqrt(ww);
                axpy3(k, w, o, o);
            }
        } break;
        case K_HULL: {
            // direction in local coordinates, first maximum over the vertex list, that vertex back to the world
            const double dl0 = dot3(d, s.ax[0]), dl1 = dot3(d, s.ax[1]), dl2 = dot3(d, s.ax[2]);
            double v0, v1, v2;
            if (s.rad < 0.0) {
                // rad < 0 (set by the pair loop of k_distances): every lane of the wave holds THIS hull, only the poses differ -- the
                // vertex list is read through the scalar cache (constant address space, uniform pointer and count) instead of sixty-four
                // identical vector loads per coordinate; same comparisons, same vertex
                typedef const __attribute__((address_space(4))) double* ConstDoubleP;
                const unsigned long long pu = __builtin_bit_cast(unsigned long long, s.h[0]);
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pu);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pu >> 32));
                const ConstDoubleP hvu = (ConstDoubleP)(((unsigned long long)hi << 32) | lo);
                const int hnu = __builtin_amdgcn_readfirstlane(hull_hn(s));
                hull_first_max(hvu, hnu, dl0, dl1, dl2, v0, v1, v2);
            } else {
                const int hn = hull_hn(s);
                hull_first_max(hull_hv(s), hn, dl0, dl1, dl2, v0, v1, v2);       // (k_narrow*: hv may be a flat address of the LDS copy)
            }
            copy3(s.c, o);
            axpy3(v0, s.ax[0], o, o);
            axpy3(v1, s.ax[1], o, o);
            axpy3(v2, s.ax[2], o, o);
        } break;
        default: {
            copy3(s.c, o);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double dj = dot3(d, s.ax[j]);
                const double sj = dj >= 0.0 ? s.h[j] : -s.h[j];
                axpy3(sj, s.ax[j], o, o);
            }
        } break;
    }
}

NBK_DEV double core_halfwidth(const Core& s, const double* n) {
    switch (s.kind) {
        case K_POINT: return 0.0;
        case K_SEG: return s.h[0] * __builtin_fabs(dot3(n, s.ax[2]));
        case K_CYL: {
            const double nu = dot3(n, s.ax[2]);
            const double r2 = NBK_FMA(-nu, nu, 1.0);
            return NBK_FMA(s.rad, nbk_sqrt(r2 > 0.0 ? r2 : 0.0), s.h[0] * __builtin_fabs(nu));
        }
        default:
            return NBK_FMA(s.h[2], __builtin_fabs(dot3(n, s.ax[2])),
                           NBK_FMA(s.h[1], __builtin_fabs(dot3(n, s.ax[1])), s.h[0] * __builtin_fabs(dot3(n, s.ax[0]))));
    }
}

// extents of a core along unit direction n about its centre: the core spans [-neg, +pos].  Symmetric kinds: both are the
// half width; hull: pos = max_k dl.v_k, neg = -min_k dl.v_k with dl = n in local coordinates.
// LDSV: the caller may hold cores whose hull vertices k_narrow* staged in LDS (their LDS byte address in `rad`, unused by hulls and 0
// everywhere else): read them with ds_read.  Only the overlap-depth CALL of the predicate instantiates it (overlap_depth_copy below), so
// the kernels that never stage -- the distance kernels above all -- keep one loop and no branch (with the branch in the shared
// routine they lost 6 % on the mesh scene).
template <bool LDSV = false>
NBK_DEV void core_extents(const Core& s, const double* n, double& neg, double& pos) {
    if (s.kind != K_HULL) { const double hw = core_halfwidth(s, n); neg = hw; pos = hw; return; }
    const double dl0 = dot3(n, s.ax[0]), dl1 = dot3(n, s.ax[1]), dl2 = dot3(n, s.ax[2]);
    const int hn = hull_hn(s);
    if (LDSV && s.rad > 0.0) hull_min_max(reinterpret_cast<LdsDoubleP>((unsigned)s.rad), hn, dl0, dl1, dl2, neg, pos);
    else hull_min_max(hull_hv(s), hn, dl0, dl1, dl2, neg, pos);
}

// ---- GJK ---------------------------------------------------------------------------------------
// The simplex lives in registers: four slots, moved with selects (no dynamically indexed arrays, so
// nothing goes to scratch).  WIT carries the support points of A and B for witness recovery.
constexpr int GJK_MAXIT = 64;
constexpr double GJK_EPS_REL = 1e-10;
constexpr double GJK_TINY2 = 1e-30;

// Every slot is its own member, picked with SX_Y / SX_A / SX_B / SX_LAM on a COMPILE-TIME slot number: with y[4][3] the
// compiler folded the select chains below into loads and stores at a lane-varying scratch address (96-168 B of private memory
// per lane, several dependent scratch round trips per iteration).
template <bool WIT>
struct Simplex {
    double y0[3], y1[3], y2[3], y3[3];
    double a0[WIT ? 3 : 1], a1[WIT ? 3 : 1], a2[WIT ? 3 : 1], a3[WIT ? 3 : 1];
    double b0[WIT ? 3 : 1], b1[WIT ? 3 : 1], b2[WIT ? 3 : 1], b3[WIT ? 3 : 1];
    double lam0, lam1, lam2, lam3;
    int n;
};
#define SX_Y(s, i) ((i) == 0 ? (s).y0 : ((i) == 1 ? (s).y1 : ((i) == 2 ? (s).y2 : (s).y3)))
#define SX_A(s, i) ((i) == 0 ? (s).a0 : ((i) == 1 ? (s).a1 : ((i) == 2 ? (s).a2 : (s).a3)))
#define SX_B(s, i) ((i) == 0 ? (s).b0 : ((i) == 1 ? (s).b1 : ((i) == 2 ? (s).b2 : (s).b3)))
#define SX_LAM(s, i) ((i) == 0 ? (s).lam0 : ((i) == 1 ? (s).lam1 : ((i) == 2 ? (s).lam2 : (s).lam3)))

// one of four VALUES by a lane-varying slot number.  The empty asm makes the operands opaque registers: without it the compiler
// folds "select of loads from consecutive members" into one load at a lane-varying address, which keeps the whole simplex in
// scratch
NBK_DEV double sx_pick(int k, double v0, double v1, double v2, double v3) {
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    return k == 0 ? v0 : (k == 1 ? v1 : (k == 2 ? v2 : v3));
}

// result of a closest-point query on a sub-simplex: up to 3 kept slots (indices into the CURRENT simplex)
struct Closest { double v[3]; int idx[3]; double lam[3]; int n; };

template <bool WIT>
NBK_DEV void sx_get(const Simplex<WIT>& s, int i, double* y) {
    // select chain (i is lane-varying)
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = sx_pick(i, s.y0[c], s.y1[c], s.y2[c], s.y3[c]);
}

NBK_DEV void closest_seg(const double* A, const double* Bp, int i0, int i1, Closest& r) {
    double ab[3];
    sub3(Bp, A, ab);
    const double t = -dot3(A, ab);
    if (t <= 0.0) { copy3(A, r.v); r.idx[0] = i0; r.lam[0] = 1.0; r.n = 1; return; }
    const double den = dot3(ab, ab);
    if (t >= den) { copy3(Bp, r.v); r.idx[0] = i1; r.lam[0] = 1.0; r.n = 1; return; }
    const double tt = t / den;
    axpy3(tt, ab, A, r.v);
    r.idx[0] = i0; r.idx[1] = i1; r.lam[0] = 1.0 - tt; r.lam[1] = tt; r.n = 2;
}

NBK_DEV void closest_tri(const double* A, const double* Bp, const double* Cp, int i0, int i1, int i2, Closest& r) {
    double ab[3], ac[3];
    sub3(Bp, A, ab); sub3(Cp, A, ac);
    const double d1 = -dot3(ab, A), d2 = -dot3(ac, A);
    if (d1 <= 0.0 && d2 <= 0.0) { copy3(A, r.v); r.idx[0] = i0; r.lam[0] = 1.0; r.n = 1; return; }
    const double d3 = -dot3(ab, Bp), d4 = -dot3(ac, Bp);
    if (d3 >= 0.0 && d4 <= d3) { copy3(Bp, r.v); r.idx[0] = i1; r.lam[0] = 1.0; r.n = 1; return; }
    const double vc = NBK_FMA(d1, d4, -(d3 * d2));
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double t = d1 / (d1 - d3);
        axpy3(t, ab, A, r.v);
        r.idx[0] = i0; r.idx[1] = i1; r.lam[0] = 1.0 - t; r.lam[1] = t; r.n = 2;
        return;
    }
    const double d5 = -dot3(ab, Cp), d6 = -dot3(ac, Cp);
    if (d6 >= 0.0 && d5 <= d6) { copy3(Cp, r.v); r.idx[0] = i2; r.lam[0] = 1.0; r.n = 1; return; }
    const double vb = NBK_FMA(d5, d2, -(d1 * d6));
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double t = d2 / (d2 - d6);
        axpy3(t, ac, A, r.v);
        r.idx[0] = i0; r.idx[1] = i2; r.lam[0] = 1.0 - t; r.lam[1] = t; r.n = 2;
        return;
    }
    const double va = NBK_FMA(d3, d6, -(d5 * d4));
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        double bc[3];
        sub3(Cp, Bp, bc);
        axpy3(t, bc, Bp, r.v);
        r.idx[0] = i1; r.idx[1] = i2; r.lam[0] = 1.0 - t; r.lam[1] = t; r.n = 2;
        return;
    }
    const double den = 1.0 / (va + vb + vc);
    const double tv = vb * den, tw = vc * den;
    double tmp[3];
    axpy3(tv, ab, A, tmp);
    axpy3(tw, ac, tmp, r.v);
    r.idx[0] = i0; r.idx[1] = i1; r.idx[2] = i2;
    r.lam[0] = (1.0 - tv) - tw; r.lam[1] = tv; r.lam[2] = tw; r.n = 3;
}

NBK_DEV bool outside_face(const double* a, const double* b, const double* c, const double* d) {
    double ab[3], ac[3], ad[3], n[3];
    sub3(b, a, ab); sub3(c, a, ac); sub3(d, a, ad);
    cross3(ab, ac, n);
    const double sp = -dot3(a, n);
    const double sd = dot3(ad, n);
    return (sp * sd < 0.0) || (sd == 0.0);
}

// keep the slots r.idx[0..n) in that order
template <bool WIT>
NBK_DEV void sx_keep(Simplex<WIT>& s, const Closest& r) {
    double ny[3][3], na[3][3], nb[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = r.idx[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ny[i][c] = sx_pick(k, s.y0[c], s.y1[c], s.y2[c], s.y3[c]);
            if constexpr (WIT) {
                na[i][c] = sx_pick(k, s.a0[c], s.a1[c], s.a2[c], s.a3[c]);
                nb[i][c] = sx_pick(k, s.b0[c], s.b1[c], s.b2[c], s.b3[c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < r.n) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                SX_Y(s, i)[c] = ny[i][c];
                if constexpr (WIT) { SX_A(s, i)[c] = na[i][c]; SX_B(s, i)[c] = nb[i][c]; }
            }
            SX_LAM(s, i) = r.lam[i];
        }
    }
    s.n = r.n;
}

// one simplex update shared by the distance and the predicate loops: place w in slot n, find the closest
// point of the enlarged simplex to the origin and shrink to its support.
// returns 0 = advanced (v, vv_prev updated), 1 = origin enclosed / touching, 2 = no progress.
// On 1 and 2 the simplex is left as it was (slots 0..n-1 and their lambdas untouched).
template <bool WIT>
NBK_DEV int gjk_advance(Simplex<WIT>& sx, const double* w, const double* sa, const double* sb, double* v, double& vv_prev) {
    const int k = sx.n;
    // (value selects into every slot, not a store under `if (i == k)`: that one becomes a store at a lane-varying address)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool here = i == k;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            SX_Y(sx, i)[c] = here ? w[c] : SX_Y(sx, i)[c];
            if constexpr (WIT) { SX_A(sx, i)[c] = here ? sa[c] : SX_A(sx, i)[c]; SX_B(sx, i)[c] = here ? sb[c] : SX_B(sx, i)[c]; }
        }
    }
    Closest r;
    r.n = 0; r.idx[0] = r.idx[1] = r.idx[2] = 0; r.lam[0] = r.lam[1] = r.lam[2] = 0.0;
    r.v[0] = r.v[1] = r.v[2] = 0.0;
    bool inside = false;
    if (k == 0) {
        copy3(sx.y0, r.v); r.idx[0] = 0; r.lam[0] = 1.0; r.n = 1;
    } else if (k == 1) {
        closest_seg(sx.y0, sx.y1, 0, 1, r);
    } else if (k == 2) {
        closest_tri(sx.y0, sx.y1, sx.y2, 0, 1, 2, r);
    } else {
        // faces (0,1,2|3) (0,2,3|1) (0,3,1|2) (1,3,2|0): best of the faces the origin is outside of
        double best = NBK_INF;
        bool any = false;
        Closest cr;
        if (outside_face(sx.y0, sx.y1, sx.y2, sx.y3)) {
            any = true;
            closest_tri(sx.y0, sx.y1, sx.y2, 0, 1, 2, cr);
            const double dd = dot3(cr.v, cr.v);
            if (dd < best) { best = dd; r = cr; }
        }
        if (outside_face(sx.y0, sx.y2, sx.y3, sx.y1)) {
            any = true;
            closest_tri(sx.y0, sx.y2, sx.y3, 0, 2, 3, cr);
            const double dd = dot3(cr.v, cr.v);
            if (dd < best) { best = dd; r = cr; }
        }
        if (outside_face(sx.y0, sx.y3, sx.y1, sx.y2)) {
            any = true;
            closest_tri(sx.y0, sx.y3, sx.y1, 0, 3, 1, cr);
            const double dd = dot3(cr.v, cr.v);
            if (dd < best) { best = dd; r = cr; }
        }
        if (outside_face(sx.y1, sx.y3, sx.y2, sx.y0)) {
            any = true;
            closest_tri(sx.y1, sx.y3, sx.y2, 1, 3, 2, cr);
            const double dd = dot3(cr.v, cr.v);
            if (dd < best) { best = dd; r = cr; }
        }
        inside = !any;
    }
    if (inside) return 1;
    const double nn = dot3(r.v, r.v);
    if (nn <= GJK_TINY2) return 1;
    if (nn >= vv_prev) return 2;
    sx_keep<WIT>(sx, r);
    vv_prev = nn;
    copy3(r.v, v);
    return 0;
}

template <bool WIT>
NBK_DEV void sx_init(Simplex<WIT>& sx) {
    sx.n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        SX_LAM(sx, i) = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) SX_Y(sx, i)[c] = 0.0;
    }
    if constexpr (WIT) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) { SX_A(sx, i)[c] = 0.0; SX_B(sx, i)[c] = 0.0; }
    }
}

template <bool WIT>
NBK_DEV bool sx_has(const Simplex<WIT>& sx, const double* w) {
    bool dup = false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < sx.n && SX_Y(sx, i)[0] == w[0] && SX_Y(sx, i)[1] == w[1] && SX_Y(sx, i)[2] == w[2]) dup = true;
    return dup;
}

// GJK distance.  returns true when the cores overlap (or touch); otherwise vout = closest vector (B -> A)
// and, with WIT, pa/pb the witness points.  `sep` = a separating plane has been proven (v.w > 0); after
// that an "enclosed" verdict can only be breakdown of a degenerate simplex and just ends the iteration.
// Convergence: |v|^2 - max_k (v_k.w_k)^2/|v_k|^2 <= eps |v|^2 (best lower bound seen so far).
template <bool WIT>
NBK_DEV bool gjk_cores(const Core& A, const Core& Bc, double* vout, double* pa, double* pb) {
    Simplex<WIT> sx;
    sx_init<WIT>(sx);
    double v[3];
    sub3(A.c, Bc.c, v);
    if (dot3(v, v) == 0.0) { v[0] = 1.0; v[1] = 0.0; v[2] = 0.0; }
    double vv_prev = NBK_INF, lb2 = 0.0;
    bool overlap = false, sep = false;
    for (int it = 0; it < GJK_MAXIT; ++it) {
        const double nv[3] = {-v[0], -v[1], -v[2]};
        double sa[3], sb[3], w[3];
        core_support(A, nv, sa);
        core_support(Bc, v, sb);
        sub3(sa, sb, w);
        const double vv = dot3(v, v);
        const double vw = dot3(v, w);
        if (vw > 0.0) {
            sep = true;
            const double l2 = (vw * vw) / vv;
            if (l2 > lb2) lb2 = l2;
        }
        if (sx.n > 0 && (vv - lb2) <= GJK_EPS_REL * vv) break;
        if (sx_has<WIT>(sx, w)) break;
        const int st = gjk_advance<WIT>(sx, w, sa, sb, v, vv_prev);
        if (st == 1) { overlap = !sep; break; }
        if (st == 2) break;
    }
    if (overlap) return true;
    copy3(v, vout);
    if constexpr (WIT) {
        pa[0] = pa[1] = pa[2] = 0.0; pb[0] = pb[1] = pb[2] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < sx.n) { axpy3(SX_LAM(sx, i), SX_A(sx, i), pa, pa); axpy3(SX_LAM(sx, i), SX_B(sx, i), pb, pb); }
        }
    }
    return false;
}

template <bool LDSV = false>
NBK_DEV double overlap_depth(const Core& A, const Core& Bc, double* normal);
// the predicate's use of it (negative tc, penetrating cores) goes through a real call on COPIES of the cores: inlined into the
// GJK loop its fifteen-axis family drove k_narrow / k_narrow_pred to 600+ spilled VGPRs once the cores lived in registers
__device__ __attribute__((noinline)) bool overlap_deeper_copy(Core A, Core Bc, double x);
NBK_DEV bool overlap_deeper_call(const Core& A, const Core& Bc, double x) { return overlap_deeper_copy(A, Bc, x); }

// GJK predicate: dist(coreA, coreB) < tc ?  Same iteration, but it returns as soon as the support-plane
// lower bound reaches tc (free) or the simplex point drops below tc (colliding).
// The predicate as a resumable state machine: gjk_pred_init once, then gjk_pred_step until it returns non-zero
// (1 = free, 2 = colliding).  One step = one GJK iteration.  k_narrow advances all its lanes one step at a time
// and hands a finished lane the next queued item, so the lanes of a wave stay busy although the iteration
// counts of the items differ (mean 2.2, max ~8 on the benchmark scene); gjk_collides below is the plain loop.
struct GjkPred {
    Simplex<false> sx;
    double v[3];
    double vv_prev, lb2;
    bool sep;
    int it;
};

NBK_DEV void gjk_pred_init(GjkPred& g, const Core& A, const Core& Bc) {
    sx_init<false>(g.sx);
    sub3(A.c, Bc.c, g.v);
    if (dot3(g.v, g.v) == 0.0) { g.v[0] = 1.0; g.v[1] = 0.0; g.v[2] = 0.0; }
    g.vv_prev = NBK_INF;
    g.lb2 = 0.0;
    g.sep = false;
    g.it = 0;
}

// POSITIVE: the caller guarantees tc > 0, which compiles the overlap-depth estimate (needed only for negative tc) out
template <bool POSITIVE = false>
NBK_DEV int gjk_pred_step(GjkPred& g, const Core& A, const Core& Bc, double tc) {
    const double tc2 = tc * tc;
    bool finish = g.it >= GJK_MAXIT;       // out of iterations: decide on the current simplex point
    if (!finish) {
        const double nv[3] = {-g.v[0], -g.v[1], -g.v[2]};
        double sa[3], sb[3], w[3];
        core_support(A, nv, sa);
        core_support(Bc, g.v, sb);
        sub3(sa, sb, w);
        const double vv = dot3(g.v, g.v);
        const double vw = dot3(g.v, w);
        if (vw > 0.0) {
            g.sep = true;
            if (tc <= 0.0) return 1;
            if (vw * vw >= tc2 * vv) return 1;
            const double l2 = (vw * vw) / vv;
            if (l2 > g.lb2) g.lb2 = l2;
        }
        if (g.sx.n > 0 && (vv - g.lb2) <= GJK_EPS_REL * vv) finish = true;
        else if (sx_has<false>(g.sx, w)) finish = true;
        else {
            const int st = gjk_advance<false>(g.sx, w, sa, sb, g.v, g.vv_prev);
            if (st == 1) {
                if (g.sep) finish = true;
                else {
                    if (POSITIVE || tc >= 0.0) return 2;
                    if constexpr (!POSITIVE) {
                        return overlap_deeper_call(A, Bc, -tc) ? 2 : 1;
                    }
                }
            } else if (st == 2) finish = true;
            else {
                if (tc > 0.0 && g.vv_prev < tc2) return 2;
                g.it += 1;
                return 0;
            }
        }
    }
    return (nbk_sqrt(dot3(g.v, g.v)) < tc) ? 2 : 1;
}

template <bool LDSV>
NBK_DEV double overlap_depth(const Core& A, const Core& Bc, double* normal);

NBK_DEV bool gjk_collides(const Core& A, const Core& Bc, double tc) {
    GjkPred g;
    gjk_pred_init(g, A, Bc);
    while (true) {
        const int r = gjk_pred_step(g, A, Bc, tc);
        if (r != 0) return r == 2;
    }
}

// ---- boolean GJK (the predicate for tc == 0): mirrors gjk_intersect of the oracle ------------------------------
constexpr int GJKB_MAXIT = 32;
constexpr int GJKB_INFL_MAXIT = 64;     // the inflated walk (tc > 0): then the distance iteration decides (figures: see the oracle)
struct GjkBool { double p[3][3]; int n; double d[3]; int it; };   // p[0] oldest; at most 3 points are kept between steps

NBK_DEV void mink_support(const Core& A, const Core& Bc, const double* d, double* w) {
    const double nd[3] = {-d[0], -d[1], -d[2]};
    double sa[3], sb[3];
    core_support(A, d, sa);
    core_support(Bc, nd, sb);
    sub3(sa, sb, w);
}
NBK_DEV void tri_prod(const double* x, const double* y, double* o) {   // (x cross y) cross x
    double t[3];
    cross3(x, y, t);
    cross3(t, x, o);
}
NBK_DEV void gjkb_triangle(GjkBool& g, const double* c_in, const double* b_in, const double* a_in) {
    const double c[3] = {c_in[0], c_in[1], c_in[2]}, b[3] = {b_in[0], b_in[1], b_in[2]}, a[3] = {a_in[0], a_in[1], a_in[2]};
    double ab[3], ac[3], abc[3], t[3];
    const double ao[3] = {-a[0], -a[1], -a[2]};
    sub3(b, a, ab); sub3(c, a, ac);
    cross3(ab, ac, abc);
    cross3(abc, ac, t);
    bool star = false;
    if (dot3(t, ao) > 0.0) {
        if (dot3(ac, ao) > 0.0) {
            copy3(c, g.p[0]); copy3(a, g.p[1]); g.n = 2;
            tri_prod(ac, ao, g.d);
            return;
        }
        star = true;
    } else {
        cross3(ab, abc, t);
        if (dot3(t, ao) > 0.0) star = true;
    }
    if (star) {
        if (dot3(ab, ao) > 0.0) { copy3(b, g.p[0]); copy3(a, g.p[1]); g.n = 2; tri_prod(ab, ao, g.d); }
        else { copy3(a, g.p[0]); g.n = 1; copy3(ao, g.d); }
        return;
    }
    if (dot3(abc, ao) > 0.0) {
        copy3(c, g.p[0]); copy3(b, g.p[1]); copy3(a, g.p[2]);
        copy3(abc, g.d);
    } else {
        copy3(b, g.p[0]); copy3(c, g.p[1]); copy3(a, g.p[2]);
        g.d[0] = -abc[0]; g.d[1] = -abc[1]; g.d[2] = -abc[2];
    }
    g.n = 3;
}
NBK_DEV void gjkb_init(GjkBool& g, const Core& A, const Core& Bc) {
    sub3(A.c, Bc.c, g.d);
    if (dot3(g.d, g.d) == 0.0) { g.d[0] = 1.0; g.d[1] = 0.0; g.d[2] = 0.0; }
    g.n = 0;
    g.it = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { g.p[i][0] = 0.0; g.p[i][1] = 0.0; g.p[i][2] = 0.0; }
}
// one iteration: 0 = continue, 1 = free, 2 = intersecting, 3 = undecided (inflated walk only).
// INFL 0: the cores as they are (tc == 0).  INFL 1: core A inflated by a ball of radius tc > 0 -- its support point moves by
// tc d/|d| -- so that "A (+) ball(tc) meets B" decides dist(A, B) < tc with the cheap walk instead of the distance iteration;
// the rounded shape can take long to separate from a near-tangent partner, so the walk gives up after GJKB_INFL_MAXIT steps
// (or with the origin on the simplex) and the caller falls back to gjk_collides.  INFL 2: per item, inflated iff tc > 0.
template <int INFL = 0>
NBK_DEV int gjkb_step(GjkBool& g, const Core& A, const Core& Bc, double tc = 0.0) {
    const bool infl = INFL == 1 || (INFL == 2 && tc > 0.0);
    if (g.it >= (infl ? GJKB_INFL_MAXIT : GJKB_MAXIT)) return infl ? 3 : 2;
    double a[3];
    mink_support(A, Bc, g.d, a);
    if (infl) {
        const double k = tc / nbk_sqrt(dot3(g.d, g.d));
        axpy3(k, g.d, a, a);
    }
    if (dot3(a, g.d) < 0.0) return 1;
    g.it += 1;
    if (g.n == 0) {
        copy3(a, g.p[0]); g.n = 1;
        g.d[0] = -a[0]; g.d[1] = -a[1]; g.d[2] = -a[2];
    } else if (g.n == 1) {
        double ab[3];
        const double ao[3] = {-a[0], -a[1], -a[2]};
        sub3(g.p[0], a, ab);
        if (dot3(ab, ao) > 0.0) { copy3(a, g.p[1]); g.n = 2; tri_prod(ab, ao, g.d); }
        else { copy3(a, g.p[0]); g.n = 1; copy3(ao, g.d); }
    } else {
        // n == 2: triangle (p0, p1, a); n == 3: the face of the tetrahedron (dd, c, b, a) that sees the origin, tested in
        // the order abc, acd, adb -- the face is picked with selects so that the triangle case is instantiated once
        double x[3], y[3];
        bool enclosed = false;
        if (g.n == 2) { copy3(g.p[0], x); copy3(g.p[1], y); }
        else {
            double ab[3], ac[3], ad[3], abc[3], acd[3], adb[3];
            const double ao[3] = {-a[0], -a[1], -a[2]};
            const double dd[3] = {g.p[0][0], g.p[0][1], g.p[0][2]}, c[3] = {g.p[1][0], g.p[1][1], g.p[1][2]}, b[3] = {g.p[2][0], g.p[2][1], g.p[2][2]};
            sub3(b, a, ab); sub3(c, a, ac); sub3(dd, a, ad);
            cross3(ab, ac, abc); cross3(ac, ad, acd); cross3(ad, ab, adb);
            const double sabc = dot3(abc, ad) > 0.0 ? -1.0 : 1.0;
            const double sacd = dot3(acd, ab) > 0.0 ? -1.0 : 1.0;
            const double sadb = dot3(adb, ac) > 0.0 ? -1.0 : 1.0;
            const bool f0 = sabc * dot3(abc, ao) > 0.0;
            const bool f1 = !f0 && sacd * dot3(acd, ao) > 0.0;
            const bool f2 = !f0 && !f1 && sadb * dot3(adb, ao) > 0.0;
            enclosed = !f0 && !f1 && !f2;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                x[e] = f0 ? c[e] : (f1 ? dd[e] : b[e]);
                y[e] = f0 ? b[e] : (f1 ? c[e] : dd[e]);
            }
        }
        if (enclosed) return 2;
        gjkb_triangle(g, x, y, a);
    }
    if (dot3(g.d, g.d) == 0.0) return infl ? 3 : 2;
    return 0;
}
NBK_DEV bool gjk_intersect(const Core& A, const Core& Bc) {
    GjkBool g;
    gjkb_init(g, A, Bc);
    while (true) { const int r = gjkb_step(g, A, Bc); if (r != 0) return r == 2; }
}
// dist(A, B) < tc for tc > 0 by the inflated walk: 1 colliding, 0 free, -1 undecided
NBK_DEV int gjk_intersect_inflated(const Core& A, const Core& Bc, double tc) {
    GjkBool g;
    gjkb_init(g, A, Bc);
    while (true) { const int r = gjkb_step<1>(g, A, Bc, tc); if (r != 0) return r == 3 ? -1 : (r == 2 ? 1 : 0); }
}

// ---- overlap depth over the candidate axis family ---------------------------------------------
template <bool LDSV = false>
NBK_DEV void try_axis(const Core& A, const Core& Bc, const double* delta, const double* n_in, double& best, double* bn) {
    const double nn = dot3(n_in, n_in);
    if (!(nn > 1e-24)) return;
    const double inv = 1.0 / nbk_sqrt(nn);
    const double n[3] = {n_in[0] * inv, n_in[1] * inv, n_in[2] * inv};
    const double proj = dot3(n, delta);
    if (A.kind == K_HULL || Bc.kind == K_HULL) {
        // not centrally symmetric: pushing A along +n separates after tp = (aN + bP) - proj, along -n after tm = (aP + bN) + proj
        double aN, aP, bN, bP;
        core_extents<LDSV>(A, n, aN, aP);
        core_extents<LDSV>(Bc, n, bN, bP);
        const double tp = (aN + bP) - proj, tm = (aP + bN) + proj;
        const double ov = tp <= tm ? tp : tm;
        if (ov < best) {
            best = ov;
            const double sg = tp <= tm ? 1.0 : -1.0;
            bn[0] = sg * n[0]; bn[1] = sg * n[1]; bn[2] = sg * n[2];
        }
        return;
    }
    const double ov = (core_halfwidth(A, n) + core_halfwidth(Bc, n)) - __builtin_fabs(proj);
    if (ov < best) {
        best = ov;
        const double sg = proj >= 0.0 ? 1.0 : -1.0;
        bn[0] = sg * n[0]; bn[1] = sg * n[1]; bn[2] = sg * n[2];
    }
}

// world direction of face f of a hull core
NBK_DEV void hull_face_normal(const Core& s, int f, double* n) {
    const double* pl = hull_hp(s) + 4 * f;
    const double p0 = pl[0], p1 = pl[1], p2 = pl[2];
    n[0] = 0.0; n[1] = 0.0; n[2] = 0.0;
    axpy3(p0, s.ax[0], n, n);
    axpy3(p1, s.ax[1], n, n);
    axpy3(p2, s.ax[2], n, n);
}

NBK_DEV int core_naxes(const Core& s) { return s.kind == K_BOX ? 3 : ((s.kind == K_SEG || s.kind == K_CYL) ? 1 : 0); }
// axis i of the family of core s: box -> ax[i]; seg/cyl -> ax[2]
NBK_DEV const double* core_axis(const Core& s, int i) { return s.kind == K_BOX ? s.ax[i] : s.ax[2]; }

template <bool LDSV>
NBK_DEV double overlap_depth(const Core& A, const Core& Bc, double* normal) {
    double delta[3];
    sub3(A.c, Bc.c, delta);
    double best = NBK_INF;
    normal[0] = 1.0; normal[1] = 0.0; normal[2] = 0.0;
    const int na = core_naxes(A), nb = core_naxes(Bc);
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i < na) try_axis<LDSV>(A, Bc, delta, core_axis(A, i), best, normal);
#pragma unroll
    for (int j = 0; j < 3; ++j) if (j < nb) try_axis<LDSV>(A, Bc, delta, core_axis(Bc, j), best, normal);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (i < na && j < nb) {
                double cr[3];
                cross3(core_axis(A, i), core_axis(Bc, j), cr);
                try_axis<LDSV>(A, Bc, delta, cr, best, normal);
            }
    // face normals of hull cores (exact for a point inside a hull, an upper bound otherwise: no edge-edge axes)
    if (A.kind == K_HULL) for (int f = 0, nf = hull_hf(A); f < nf; ++f) { double fn[3]; hull_face_normal(A, f, fn); try_axis<LDSV>(A, Bc, delta, fn, best, normal); }
    if (Bc.kind == K_HULL) for (int f = 0, nf = hull_hf(Bc); f < nf; ++f) { double fn[3]; hull_face_normal(Bc, f, fn); try_axis<LDSV>(A, Bc, delta, fn, best, normal); }
    if (A.kind == K_CYL) { double r[3]; axpy3(-dot3(delta, A.ax[2]), A.ax[2], delta, r); try_axis<LDSV>(A, Bc, delta, r, best, normal); }
    if (Bc.kind == K_CYL) { double r[3]; axpy3(-dot3(delta, Bc.ax[2]), Bc.ax[2], delta, r); try_axis<LDSV>(A, Bc, delta, r, best, normal); }
    try_axis<LDSV>(A, Bc, delta, delta, best, normal);
    if (best == NBK_INF) best = 0.0;
    return best;
}

__device__ __attribute__((noinline)) int epa_depth_copy(Core A, Core Bc, double* out, int decide, double x);
// is the exact depth of two overlapping cores larger than x (= -tc > 0: the predicate at a negative contact threshold)?  The family's
// value is an upper bound (not deeper: done) and exact without a cylinder / hull core; EPA stops at its first certain answer
// (oracle: overlap_deeper_than)
__device__ __attribute__((noinline)) bool overlap_deeper_copy(Core A, Core Bc, double x) {
    double nrm[3];
    // a hull's family scans faces x vertices: there EPA goes first and the family is consulted only when EPA leaves the question open;
    // everywhere else the family is a handful of axes and settles most items before a polytope is built
    const bool hull = A.kind == K_HULL || Bc.kind == K_HULL;
    if (!hull) {
        const double fam = overlap_depth<false>(A, Bc, nrm);
        if (!(fam > x)) return false;
        if (!(A.kind == K_CYL || Bc.kind == K_CYL)) return true;
    }
    // (hull cores staged in LDS keep their LDS byte address in `rad`: the support routine ignores it)
    double o[4];
    const int r = epa_depth_copy(A, Bc, o, 1, x);
    if (r == 2) return true;
    if (r == 3) return false;
    if (hull) {
        double fam;
        const bool lds_a = A.rad > 0.0 && A.kind == K_HULL, lds_b = Bc.rad > 0.0 && Bc.kind == K_HULL;
        if (lds_a || lds_b) fam = overlap_depth<true>(A, Bc, nrm);
        else fam = overlap_depth<false>(A, Bc, nrm);
        if (!(fam > x)) return false;
    }
    if (r == 1) return o[0] > x;
    return true;                       // no answer from EPA: the family's value stands
}

// ---- exact penetration depth of overlapping cores with a cylinder or a hull: EPA (mirrors epa_depth of the oracle line by line:
// same operations in the same order, so the same bits).  The polytope lives in per-lane arrays (scratch): only the kernels that can
// meet overlapping cores in a distance or in a negative-threshold predicate carry it, through ONE out-of-line call.
#define EPA_MAXIT 32
#define EPA_TOL 1e-8     /* relative gap between the inner polytope and the body along the nearest face's normal (Bullet's own EPA stops at 1e-4) */
#define EPA_MAXV (4 + EPA_MAXIT)
#define EPA_MAXF (4 + 2 * EPA_MAXIT + 8)
#define EPA_MAXE 64
// The polytope lives in per-lane scratch (4 KB), where a dependent load costs most of a microsecond: which faces are alive is a
// bit mask in registers (no flag loads, the lowest free slot is a bit scan), a face's three vertex indices are one packed word, the
// scans over the faces load unconditionally so that the unrolled trips overlap their loads, and the horizon is built from a
// visibility mask in ascending face order -- the order, the arithmetic and therefore every result are those of the oracle's loops.
struct Epa { double v[EPA_MAXV][3]; double fn[EPA_MAXF][3]; double fd[EPA_MAXF]; int f[EPA_MAXF]; double ref[3]; int nv, nf;
             unsigned long long alive0; unsigned alive1; };
static_assert(EPA_MAXF <= 96 && EPA_MAXV <= 255, "alive masks / packed vertex indices");
NBK_DEV bool epa_alive(const Epa& e, int q) { return q < 64 ? ((e.alive0 >> q) & 1ull) != 0ull : ((e.alive1 >> (q - 64)) & 1u) != 0u; }
NBK_DEV void epa_set_alive(Epa& e, int q, bool on) {
    if (q < 64) { const unsigned long long b = 1ull << q; e.alive0 = on ? (e.alive0 | b) : (e.alive0 & ~b); }
    else { const unsigned b = 1u << (q - 64); e.alive1 = on ? (e.alive1 | b) : (e.alive1 & ~b); }
}

NBK_DEV bool epa_plane_of(const double* vi, const double* vj, const double* vk, double* n, double& d) {
    double ab[3], ac[3], c[3];
    sub3(vj, vi, ab); sub3(vk, vi, ac);
    cross3(ab, ac, c);
    const double cc = dot3(c, c);
    if (!(cc > 1e-60)) return false;
    const double inv = 1.0 / nbk_sqrt(cc);
    n[0] = c[0] * inv; n[1] = c[1] * inv; n[2] = c[2] * inv;
    d = dot3(n, vi);
    return true;
}
NBK_DEV bool epa_face_plane(const Epa& e, int i, int j, int k, double* n, double& d) { return epa_plane_of(e.v[i], e.v[j], e.v[k], n, d); }
NBK_DEV bool epa_add_face(Epa& e, int i, int j, int k) {
    double n[3], d, r[3];
    if (!epa_face_plane(e, i, j, k, n, d)) return false;
    sub3(e.v[i], e.ref, r);
    if (dot3(n, r) < 0.0) { const int t = j; j = k; k = t; d = -d; n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    // the lowest free slot below nf, else a new one
    int slot = -1;
    {
        const unsigned long long m0 = e.nf >= 64 ? ~0ull : ((1ull << e.nf) - 1ull);
        const unsigned long long free0 = ~e.alive0 & m0;
        if (free0 != 0ull) slot = __builtin_ctzll(free0);
        else if (e.nf > 64) {
            const unsigned free1 = ~e.alive1 & ((1u << (e.nf - 64)) - 1u);
            if (free1 != 0u) slot = 64 + __builtin_ctz(free1);
        }
    }
    if (slot < 0) { if (e.nf >= EPA_MAXF) return false; slot = e.nf++; }
    epa_set_alive(e, slot, true);
    e.f[slot] = i | (j << 8) | (k << 16); e.fd[slot] = d;
    e.fn[slot][0] = n[0]; e.fn[slot][1] = n[1]; e.fn[slot][2] = n[2];
    return true;
}
// 1: out[0] = depth, out[1..3] = direction from B to A; 0: no answer (the caller keeps the axis-family value); with `decide` also
// 2: certainly deeper than x (the nearest face of the inner polytope is farther than x), 3: certainly not (a support plane within x)
__device__ __attribute__((noinline)) int epa_depth_copy(Core A, Core Bc, double* out, int decide, double x) {
    double sv[4][3];                  // the start tetrahedron, in registers until the polytope is needed
    {
        const double D0[3] = {0.5345224838248488, -0.2672612419124244, 0.8017837257372732};
        const double nD0[3] = {-D0[0], -D0[1], -D0[2]};
        mink_support(A, Bc, D0, sv[0]);
        mink_support(A, Bc, nD0, sv[1]);
        double e1[3];
        sub3(sv[1], sv[0], e1);
        const double l1 = dot3(e1, e1);
        if (!(l1 > 1e-30)) return 0;
        const double ax = __builtin_fabs(e1[0]), ay = __builtin_fabs(e1[1]), az = __builtin_fabs(e1[2]);
        double a[3] = {0.0, 0.0, 0.0};
        if (ax <= ay && ax <= az) a[0] = 1.0; else if (ay <= az) a[1] = 1.0; else a[2] = 1.0;
        double n2[3], nn2[3], pa2[3], pb2[3], ca[3], cb[3], ra[3], rb[3];
        axpy3(-dot3(a, e1) / l1, e1, a, n2);
        nn2[0] = -n2[0]; nn2[1] = -n2[1]; nn2[2] = -n2[2];
        mink_support(A, Bc, n2, pa2);
        mink_support(A, Bc, nn2, pb2);
        sub3(pa2, sv[0], ra); sub3(pb2, sv[0], rb);
        cross3(e1, ra, ca); cross3(e1, rb, cb);
        const bool use_a = dot3(ca, ca) >= dot3(cb, cb);
        copy3(use_a ? pa2 : pb2, sv[2]);
        double n3[3], nn3[3], pa3[3], pb3[3];
        copy3(use_a ? ca : cb, n3);
        const double l3 = dot3(n3, n3);
        if (!(l3 > 1e-24 * l1 * l1)) return 0;
        nn3[0] = -n3[0]; nn3[1] = -n3[1]; nn3[2] = -n3[2];
        mink_support(A, Bc, n3, pa3);
        mink_support(A, Bc, nn3, pb3);
        sub3(pa3, sv[0], ra); sub3(pb3, sv[0], rb);
        const double ha = __builtin_fabs(dot3(n3, ra)), hb = __builtin_fabs(dot3(n3, rb));
        copy3(ha >= hb ? pa3 : pb3, sv[3]);
        const double hh = ha >= hb ? ha : hb;
        if (!(hh * hh > 1e-24 * l3 * l1)) return 0;
    }
    Epa e;
#pragma unroll
    for (int i = 0; i < 4; ++i) { e.v[i][0] = sv[i][0]; e.v[i][1] = sv[i][1]; e.v[i][2] = sv[i][2]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) e.ref[c] = 0.25 * (((sv[0][c] + sv[1][c]) + sv[2][c]) + sv[3][c]);
    e.nv = 4; e.nf = 0; e.alive0 = 0ull; e.alive1 = 0u;
    if (!epa_add_face(e, 0, 1, 2) || !epa_add_face(e, 0, 1, 3) || !epa_add_face(e, 0, 2, 3) || !epa_add_face(e, 1, 2, 3)) return 0;
    double best_up = NBK_INF, best_n[3] = {1.0, 0.0, 0.0};
    for (int it = 0; it < EPA_MAXIT; ++it) {
        // the alive face with the smallest plane distance, the first of equals
        int bf = -1;
        double bd = 0.0;
#pragma unroll 4
        for (int q = 0; q < e.nf; ++q) {
            const double dq = e.fd[q];
            if (epa_alive(e, q) && (bf < 0 || dq < bd)) { bf = q; bd = dq; }
        }
        if (bf < 0) break;
        const double n[3] = {e.fn[bf][0], e.fn[bf][1], e.fn[bf][2]};
        const double d = bd;
        if (decide && d > x) return 2;
        double w[3];
        mink_support(A, Bc, n, w);
        const double dw = dot3(n, w);
        if (dw < best_up) { best_up = dw; best_n[0] = n[0]; best_n[1] = n[1]; best_n[2] = n[2]; }
        if (decide && best_up <= x) return 3;
        if (dw - d <= EPA_TOL * (1.0 + __builtin_fabs(dw))) break;
        if (e.nv >= EPA_MAXV) break;
        const int wi = e.nv++;
        copy3(w, e.v[wi]);
        // the faces that see w
        unsigned long long vis0 = 0ull;
        unsigned vis1 = 0u;
#pragma unroll 2
        for (int q = 0; q < e.nf; ++q) {
            const double fq[3] = {e.fn[q][0], e.fn[q][1], e.fn[q][2]};
            const double sq = dot3(fq, w) - e.fd[q];
            if (epa_alive(e, q) && !(sq <= 0.0)) { if (q < 64) vis0 |= 1ull << q; else vis1 |= 1u << (q - 64); }
        }
        e.alive0 &= ~vis0; e.alive1 &= ~vis1;
        // their edges, in ascending face order; an edge met twice is interior
        int edges[EPA_MAXE];
        int ne = 0;
        bool overflow = false;
        while (vis0 != 0ull || vis1 != 0u) {
            int q;
            if (vis0 != 0ull) { q = __builtin_ctzll(vis0); vis0 &= vis0 - 1ull; }
            else { q = 64 + __builtin_ctz(vis1); vis1 &= vis1 - 1u; }
            const int fw = e.f[q];
            const int fv[3] = {fw & 255, (fw >> 8) & 255, (fw >> 16) & 255};
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int a = fv[s3], b = fv[(s3 + 1) % 3];
                const int ab = (a << 8) | b, ba = (b << 8) | a;
                int found = -1;
                for (int t = 0; t < ne; ++t) { const int et = edges[t]; if (et == ba || et == ab) { found = t; break; } }
                if (found >= 0) { edges[found] = edges[ne - 1]; --ne; }
                else if (ne < EPA_MAXE) { edges[ne] = ab; ++ne; }
                else overflow = true;
            }
        }
        if (ne < 3 || overflow) break;
        bool bad = false;
        for (int t = 0; t < ne; ++t) { const int et = edges[t]; if (!epa_add_face(e, et >> 8, et & 255, wi)) { bad = true; break; } }
        if (bad) break;
    }
    if (!(best_up < NBK_INF)) return 0;
    out[0] = best_up > 0.0 ? best_up : 0.0;
    out[1] = -best_n[0]; out[2] = -best_n[1]; out[3] = -best_n[2];
    return 1;
}
// depth and direction (from B to A) of two overlapping cores: the axis family, tightened by EPA where the family is only a bound
template <bool LDSV = false>
NBK_DEV double overlap_depth_exact(const Core& A, const Core& Bc, double* normal) {
    double depth = overlap_depth<LDSV>(A, Bc, normal);
    if (A.kind == K_CYL || A.kind == K_HULL || Bc.kind == K_CYL || Bc.kind == K_HULL) {
        double o[4];
        if (epa_depth_copy(A, Bc, o, 0, 0.0) == 1 && o[0] < depth) { depth = o[0]; normal[0] = o[1]; normal[1] = o[2]; normal[2] = o[3]; }
    }
    return depth;
}

// ---- closed forms for point / segment cores ---------------------------------------------------
NBK_DEV void seg_seg_closest(const Core& A, const Core& Bc, double* pa, double* pb) {
    const double ha = A.h[0], hb = Bc.h[0];
    double r[3];
    sub3(A.c, Bc.c, r);
    const double b = dot3(A.ax[2], Bc.ax[2]), c = dot3(A.ax[2], r), f = dot3(Bc.ax[2], r);
    const double den = NBK_FMA(-b, b, 1.0);
    double s = 0.0;
    if (den > 1e-14) s = clampd(NBK_FMA(b, f, -c) / den, -ha, ha);
    double t = NBK_FMA(b, s, f);
    if (t < -hb) { t = -hb; s = clampd(NBK_FMA(b, t, -c), -ha, ha); }
    else if (t > hb) { t = hb; s = clampd(NBK_FMA(b, t, -c), -ha, ha); }
    axpy3(s, A.ax[2], A.c, pa);
    axpy3(t, Bc.ax[2], Bc.c, pb);
}

NBK_DEV void ps_closest(const Core& A, const Core& Bc, double* pa, double* pb) {
    if (A.kind == K_POINT && Bc.kind == K_POINT) {
        copy3(A.c, pa); copy3(Bc.c, pb);
    } else if (A.kind == K_POINT) {
        double d[3];
        sub3(A.c, Bc.c, d);
        const double t = clampd(dot3(d, Bc.ax[2]), -Bc.h[0], Bc.h[0]);
        copy3(A.c, pa);
        axpy3(t, Bc.ax[2], Bc.c, pb);
    } else if (Bc.kind == K_POINT) {
        double d[3];
        sub3(Bc.c, A.c, d);
        const double t = clampd(dot3(d, A.ax[2]), -A.h[0], A.h[0]);
        axpy3(t, A.ax[2], A.c, pa);
        copy3(Bc.c, pb);
    } else {
        seg_seg_closest(A, Bc, pa, pb);
    }
}

// point p against a box / cylinder core: signed core distance, closest surface point, outward normal
NBK_DEV double point_solid(const double* p, const Core& S, double* cp, double* nrm) {
    double d[3];
    sub3(p, S.c, d);
    if (S.kind == K_BOX) {
        double x[3], qx[3];
        bool outside = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            x[j] = dot3(d, S.ax[j]);
            qx[j] = clampd(x[j], -S.h[j], S.h[j]);
            if (__builtin_fabs(x[j]) > S.h[j]) outside = true;
        }
        if (outside) {
            copy3(S.c, cp);
#pragma unroll
            for (int j = 0; j < 3; ++j) axpy3(qx[j], S.ax[j], cp, cp);
            double e[3];
            sub3(p, cp, e);
            const double dist = nbk_sqrt(dot3(e, e));
            const double inv = 1.0 / dist;
            nrm[0] = e[0] * inv; nrm[1] = e[1] * inv; nrm[2] = e[2] * inv;
            return dist;
        }
        const double g0 = S.h[0] - __builtin_fabs(x[0]), g1 = S.h[1] - __builtin_fabs(x[1]), g2 = S.h[2] - __builtin_fabs(x[2]);
        int jm = 0;
        double best = g0;
        if (g1 < best) { best = g1; jm = 1; }
        if (g2 < best) { best = g2; jm = 2; }
        const double xm = jm == 0 ? x[0] : (jm == 1 ? x[1] : x[2]);
        const double sg = xm >= 0.0 ? 1.0 : -1.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double a = jm == 0 ? S.ax[0][c] : (jm == 1 ? S.ax[1][c] : S.ax[2][c]);
            nrm[c] = sg * a;
        }
        axpy3(best, nrm, p, cp);
        return -best;
    }
    const double* u = S.ax[2];
    const double z = dot3(d, u);
    double w[3];
    axpy3(-z, u, d, w);
    const double rho = nbk_sqrt(dot3(w, w));
    const double dz = __builtin_fabs(z) - S.h[0], dr = rho - S.rad;
    const double sz = z >= 0.0 ? 1.0 : -1.0;
    double rdir[3];
    if (rho > 0.0) { const double inv = 1.0 / rho; rdir[0] = w[0] * inv; rdir[1] = w[1] * inv; rdir[2] = w[2] * inv; }
    else { copy3(S.ax[0], rdir); }
    if (dz <= 0.0 && dr <= 0.0) {
        if (dr > dz) { copy3(rdir, nrm); axpy3(-dr, nrm, p, cp); return dr; }
        nrm[0] = sz * u[0]; nrm[1] = sz * u[1]; nrm[2] = sz * u[2];
        axpy3(-dz, nrm, p, cp);
        return dz;
    }
    const double zc = clampd(z, -S.h[0], S.h[0]);
    const double rc = rho < S.rad ? rho : S.rad;
    double tmp[3];
    axpy3(zc, u, S.c, tmp);
    axpy3(rc, rdir, tmp, cp);
    double e[3];
    sub3(p, cp, e);
    const double dist = nbk_sqrt(dot3(e, e));
    const double inv = 1.0 / dist;
    nrm[0] = e[0] * inv; nrm[1] = e[1] * inv; nrm[2] = e[2] * inv;
    return dist;
}

// signed distance between two shapes; wit (WIT only) = point on A, point on B, unit normal B -> A
// DEFER: where the exact depth of overlapping cores needs EPA (a cylinder or a hull core), return the axis-family value and its
// depth in *defer_depth (else left untouched): the caller runs EPA for such items later, one per lane on full waves -- inline,
// one lane would walk its polytope while 63 wait (k_distances)
template <bool WIT, bool DEFER = false>
NBK_DEV double cores_distance(const Core& A, const Core& Bc, double* wit, double* defer_depth = nullptr) {
    double pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0}, n[3] = {1, 0, 0};
    double dc;
    if (Bc.kind == K_PLANE) {
        const double* nn = Bc.ax[2];
        double d[3];
        sub3(A.c, Bc.c, d);
        const double hc = dot3(d, nn);
        double hw, hpos;
        core_extents(A, nn, hw, hpos);         // how far the core reaches below its centre
        const double dist = (hc - hw) - A.margin;
        if constexpr (WIT) {
            const double neg[3] = {-nn[0], -nn[1], -nn[2]};
            core_support(A, neg, pa);
            axpy3(-A.margin, nn, pa, pa);
            axpy3(-dist, nn, pa, pb);
            copy3(pa, wit); copy3(pb, wit + 3); copy3(nn, wit + 6);
        }
        return dist;
    }
    const bool a_ps = (A.kind == K_POINT || A.kind == K_SEG), b_ps = (Bc.kind == K_POINT || Bc.kind == K_SEG);
    if (a_ps && b_ps) {
        ps_closest(A, Bc, pa, pb);
        double e[3];
        sub3(pa, pb, e);
        dc = nbk_sqrt(dot3(e, e));
        if constexpr (WIT) {
            if (dc > 0.0) { const double inv = 1.0 / dc; n[0] = e[0] * inv; n[1] = e[1] * inv; n[2] = e[2] * inv; }
            else {
                double cr[3];
                cross3(A.ax[2], Bc.ax[2], cr);
                const double cc = dot3(cr, cr);
                if (A.kind == K_SEG && Bc.kind == K_SEG && cc > 1e-24) { const double inv = 1.0 / nbk_sqrt(cc); n[0] = cr[0] * inv; n[1] = cr[1] * inv; n[2] = cr[2] * inv; }
                else { n[0] = 1.0; n[1] = 0.0; n[2] = 0.0; }
            }
        }
    } else if (A.kind == K_POINT && Bc.kind != K_HULL) {
        double nb[3];
        dc = point_solid(A.c, Bc, pb, nb);
        copy3(A.c, pa); copy3(nb, n);
    } else if (Bc.kind == K_POINT && A.kind != K_HULL) {
        double na[3];
        dc = point_solid(Bc.c, A, pa, na);
        copy3(Bc.c, pb);
        n[0] = -na[0]; n[1] = -na[1]; n[2] = -na[2];
    } else {
        double v[3];
        const bool ov = gjk_cores<WIT>(A, Bc, v, pa, pb);
        if (!ov) {
            // the distance is the norm of GJK's own closest vector; pa/pb only serve as witnesses
            dc = nbk_sqrt(dot3(v, v));
            if constexpr (WIT) {
                const double inv = 1.0 / dc;
                n[0] = v[0] * inv; n[1] = v[1] * inv; n[2] = v[2] * inv;
            }
        } else {
            double depth;
            if constexpr (DEFER) {
                depth = overlap_depth(A, Bc, n);
                if (A.kind == K_CYL || A.kind == K_HULL || Bc.kind == K_CYL || Bc.kind == K_HULL) *defer_depth = depth;
            } else {
                depth = overlap_depth_exact(A, Bc, n);
            }
            dc = -depth;
            if constexpr (WIT) {
                const double neg[3] = {-n[0], -n[1], -n[2]};
                core_support(A, neg, pa);
                axpy3(dc, n, pa, pb);
            }
        }
    }
    const double dist = (dc - A.margin) - Bc.margin;
    if constexpr (WIT) {
        double wa[3], wb[3];
        axpy3(-A.margin, n, pa, wa);
        axpy3(Bc.margin, n, pb, wb);
        copy3(wa, wit); copy3(wb, wit + 3); copy3(n, wit + 6);
    }
    return dist;
}

// ---- validity predicate of one pair (spec = oracle cores_collide) ----------------------------------
//   1. decided on the CORE distance against tc = (thr + mA) + mB  (user order of the pair);
//   2. broadphase: |cA - cB|^2 >= ((tc + rhoA) + rhoB)^2, or a non-positive sum  => free;
//   3. exact test with the cores in canonical order (kind ascending);
//   planes (always second): t = thr + mA, hc = n.(cA - p0); hc - rhoA >= t => free, else hc - halfwidth < t.
// centre c (bounding radius rho) of the other core against the exact box bx.
// returns 0 = free, 1 = colliding, -1 = undecided.  Mirrors step 4 of the oracle's cores_collide.
NBK_DEV int box_midphase(const double* c, double rho, const Core& bx, double tc) {
    double d[3], ax[3], ex[3];
    bool inside = true;
    sub3(c, bx.c, d);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ax[j] = __builtin_fabs(dot3(d, bx.ax[j]));
        ex[j] = ax[j] - bx.h[j];
        if (ex[j] > 0.0) inside = false; else ex[j] = 0.0;
    }
    const double d2 = NBK_FMA(ex[2], ex[2], NBK_FMA(ex[1], ex[1], ex[0] * ex[0]));
    if (!inside) {
        if (tc >= 0.0) { const double r = tc + rho; if (d2 >= r * r) return 0; }
        if (tc > 0.0 && d2 < tc * tc) return 1;
        return -1;
    }
    double g = bx.h[0] - ax[0];
    if (bx.h[1] - ax[1] < g) g = bx.h[1] - ax[1];
    if (bx.h[2] - ax[2] < g) g = bx.h[2] - ax[2];
    return (-g < tc) ? 1 : -1;
}

NBK_DEV bool plane_collides(const Core& A, const Core& Pl, double thr, double rhoA) {
    double d[3];
    sub3(A.c, Pl.c, d);
    const double hc = dot3(d, Pl.ax[2]);
    const double t = thr + A.margin;
    if ((hc - rhoA) >= t) return false;            // broadphase: bounding sphere above the plane
    double hw, hpos;
    core_extents(A, Pl.ax[2], hw, hpos);
    return (hc - hw) < t;
}

// Cull for hull cores (device only; it never changes a verdict): the other core lies in the ball (c, rho), the hull inside its local
// bounding box, so a centre farther than tc + rho from that box means a core distance of at least tc -- free.  The comparison keeps a
// 1e-9 relative margin so that it can never contradict what the GJK iteration of the oracle decides in its last digits.
NBK_DEV bool hull_box_far(const double* c, double rho, const Core& H, double tc) {
    const double* ob = hull_hv(H) - 6;
    double d[3];
    sub3(c, H.c, d);
    double d2 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double ex = __builtin_fabs(dot3(d, H.ax[j]) - ob[j]) - ob[3 + j];
        if (ex > 0.0) d2 = NBK_FMA(ex, ex, d2);
    }
    const double r0 = (tc > 0.0 ? tc : 0.0) + rho;      // a non-positive threshold: disjoint is enough
    const double r = NBK_FMA(r0, 1e-9, r0);
    return d2 >= r * r;
}

// The same cull for box cores at NEGATIVE thresholds (device only; for tc >= 0 the predicate's own midphase does it, bit for bit like
// the oracle): the other core's centre farther than its bounding radius from the box => the cores are disjoint => free.
NBK_DEV bool box_far_negative(const double* c, double rho, const Core& bx) {
    double d[3];
    sub3(c, bx.c, d);
    double d2 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double ex = __builtin_fabs(dot3(d, bx.ax[j])) - bx.h[j];
        if (ex > 0.0) d2 = NBK_FMA(ex, ex, d2);
    }
    const double r = NBK_FMA(rho, 1e-9, rho);
    return d2 >= r * r && d2 > 0.0;
}

// steps 4-5 of the predicate up to (not including) GJK: 0 = free, 1 = colliding, -1 = the GJK predicate decides.
// A/Bc already in canonical order, neither is a plane.
NBK_DEV int cores_collide_pre(const Core& A, const Core& Bc, double tc) {
    if (A.kind == K_HULL && hull_box_far(Bc.c, Bc.rho, A, tc)) return 0;
    if (Bc.kind == K_HULL && hull_box_far(A.c, A.rho, Bc, tc)) return 0;
    if (tc < 0.0) {
        if (A.kind == K_BOX && box_far_negative(Bc.c, Bc.rho, A)) return 0;
        if (Bc.kind == K_BOX && box_far_negative(A.c, A.rho, Bc)) return 0;
    }
    // midphase for box cores: the other core's centre against the exact box (no square roots)
    if (A.kind == K_BOX || Bc.kind == K_BOX) {
        const bool b_is_box = Bc.kind == K_BOX;
        int verdict;
        if (b_is_box) verdict = box_midphase(A.c, A.rho, Bc, tc);
        else verdict = box_midphase(Bc.c, Bc.rho, A, tc);
        if (verdict >= 0) return verdict;
    }
    const bool a_ps = (A.kind == K_POINT || A.kind == K_SEG), b_ps = (Bc.kind == K_POINT || Bc.kind == K_SEG);
    if (a_ps && b_ps) {
        double pa[3], pb[3], e[3];
        ps_closest(A, Bc, pa, pb);
        sub3(pa, pb, e);
        return (nbk_sqrt(dot3(e, e)) < tc) ? 1 : 0;
    }
    if (A.kind == K_POINT && Bc.kind != K_HULL) { double cp[3], nb[3]; return (point_solid(A.c, Bc, cp, nb) < tc) ? 1 : 0; }
    return -1;
}

NBK_DEV bool cores_collide_exact(const Core& A, const Core& Bc, double tc) {
    const int pre = cores_collide_pre(A, Bc, tc);
    if (pre >= 0) return pre != 0;
    if (A.kind != K_HULL && Bc.kind != K_HULL) {
        // (hull cores go straight to the distance iteration: its early exits halve the number of vertex-list scans)
        if (tc == 0.0) return gjk_intersect(A, Bc);   // pure intersection test: the boolean walk
        if (tc > 0.0) {                               // the same walk on the inflated core; the distance iteration if it gives up
            const int r = gjk_intersect_inflated(A, Bc, tc);
            if (r >= 0) return r != 0;
        }
    }
    return gjk_collides(A, Bc, tc);
}

}  // namespace nbk
