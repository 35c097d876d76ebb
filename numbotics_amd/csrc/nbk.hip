// nbk.hip -- gfx950 kernels + the C-ABI of include/nbk.h.
//
// Mapping: ONE CONFIGURATION PER LANE, one 64-lane wave per workgroup.
//   * q rows are fetched with coalesced 16-byte loads into LDS and read back transposed, outputs
//     (poses, Jacobians) go through an LDS transpose so that every global access is a contiguous
//     1 KiB-per-instruction stream;
//   * the robot/scene descriptor is wave-uniform: it is read through scalar loads, and every branch on a
//     joint type / shape kind / pair index is a scalar branch;
//   * the collision kernels sweep the kinematic tree once, keep the current frame in VGPRs and park the
//     world-frame core of every robot shape in LDS as [component][lane] rows (conflict-free ds_read_b64);
//     pairs are then evaluated from LDS + scalar constants;
//   * the validity bit mask is one __ballot per wave = one 64-bit word per workgroup.
// No MFMA: this is a transform chain + branchy narrowphase, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <algorithm>
#include <vector>

#include "../../include/nbk.h"
#include "nbk_device.hpp"

namespace nbk {

// ---- device-resident descriptor -----------------------------------------------------------------
struct DevModel {
    int n_q, n_joints, n_rshapes, n_wshapes, n_pairs;
    int shape_rows;   // LDS rows (of 64 doubles) holding robot shape cores
    int frame_slots;  // saved frames (tree branches)
    const int* joint_type;        // [J]
    const int* joint_qidx;        // [J]
    const int* joint_load;        // [J] -2: parent is the frame in registers, -1: base, >=0: saved slot
    const int* joint_save;        // [J] slot to park this frame in, or -1
    const int* joint_shape_begin; // [J+2] robot shapes (in `order`) of frame k-1 are [begin[k], begin[k+1])
    const double* joint_rot;      // [J][27]
    const int* joint_kind;        // [J] 0/1/2: revolute about that coordinate axis of the joint frame (known zeros in joint_rot), 3: general revolute, 4: prismatic
    const double* joint_trans;    // [J][3]
    const double* joint_slide;    // [J][3]
    const double* joint_axis;     // [J][3]
    const double* joint_pk;       // [J][22] axis-aligned joints: (M2[r][U], M2[r][V]) r = 0..2 | (M1[r][U], M1[r][V]) | M0[r][KZ] | trans | axis | pad -- what
                                  //         joint_apply_axis reads, back to back (one prefetchable block per joint: k_jacobian_reg)
    const double* base_pose;      // [12]
    const int* rs_kind;           // [S] core kind, in frame order
    const int* rs_row;            // [S] first LDS row
    const double* rs_local;       // [S][12]
    const double* rs_core;        // [S][6] h0 h1 h2 rad margin rho
    const int* ws_kind;           // [W]
    const double* ws_core;        // [W][18] c(3) ax(9: ax0 ax1 ax2) h(3) rad margin rho
    const double* hull_blob;      // every hull's [bounding box (6) | vertices (3 n)], back to back: what the HullRef.hv pointers point into
    int hull_blob_n;              // ... in doubles; k_narrow* stage it in LDS when it is at most HULL_LDS_MAX bytes
    const int* pair_a;            // [P] index into rs_* (frame order)
    const int* pair_b;            // [P] < S robot (frame order), else S + world
    const int* pair_user;         // [P] index of this pair in the caller's pair list
    // validity tables: pairs sorted by class (0 plane, 1 closed form, 2 GJK); shape refs: >= 0 robot shape
    // (frame order), < 0 world shape ~ref
    const int* vp_tab;            // [P][4] refA, refB (user order), class, pad
    const int* vp_canon;          // [P][2] the two refs in canonical (kind-ascending) order
    const double* vp_cst;         // [P][4] mA, mB, rhoA, rhoB (user order)
    const double* ws_center;      // [W][3]
    int n_plane_pairs, n_closed_pairs;   // class boundaries inside the sorted tables
    const int* rs_frame;          // [S] moving frame of each robot shape (frame order, non-decreasing)
    const unsigned* rs_mask;      // [S] joints on the path from the base to the shape's frame (bit k = joint k)
    // float copies for the conservative float32 broadphase (k_broad_f32): [J][27] rot | [J][3] trans | [J][3] slide |
    // base[12] | [S][3] shape centre offsets | [W][18] world cores; slack = f_eps * max(f_reach, largest |coordinate| of
    // the configuration) covers the float32 error of the sweep 50 times over
    const float* f_tab;
    int f_trans, f_slide, f_base, f_tl, f_wc, f_wobb;      // f_wobb: [W][6] local bounding boxes of world hulls (centre, half extents)
    int f_pk;                     // [J][20] per-joint constants of the packed float32 sweep (JPk), 16-byte aligned
    int f_meta;                   // [8] dwords (bit patterns): joint kind | q column << 8 of the first 8 joints
    int f_chain;                  // 1: a serial chain of at most 8 joints without saved frames -- k_broad_f32 unrolls its joints at compile time
    float f_eps, f_reach;
    float f_e2max;                // static bound of the per-lane slack 2e: lanes above it (prismatic travel, |q| sums beyond 64 rad) do not certify hits
    const double* rs_in;          // [S] radius of a ball around the shape's centre that lies inside the shape (margin included)
    const double* ws_in;          // [W] the same for world shapes (0 for planes)
    const double* bq_static;      // [P] (broadphase order) static lower bound of the pair's centre distance (plane: height) minus the
                                  //     bounding radii, over ALL configurations: reach of the shape's centre from the base; -inf when unknown
    const int* vp_cls;            // [P] kind class of the pair (0 box-box, 1 box-cylinder, 2 cylinder-cylinder, 3 the rest): queue routing
    int cls_base[4], cls_groups[4];   // sub-queues [cls_base[c], cls_base[c] + cls_groups[c]) serve class c, in proportion to its pairs
    const int4* vp_info;          // [P] canonical refs and their joint masks in one 16-byte record: ra, rb, mask(ra), mask(rb)
    int bq_count[4];              // pairs per broadphase category
    const int* bq_tab;            // [P][4] broadphase order (category-major): centre row of A (3*shape), centre row of B or world index,
                                  //        index into the vp_* tables, category (0 plane, 1 robot-robot, 2 robot-world, 3 robot-world box)
    int dbg;                              // ablation switches, ONLY in builds made with -DNBK_ABLATE_BUILD (tools/ablate.py, tools/narrow_prof.py;
                                          // the product library compiles them out: NBK_DBG is the constant 0): 1 = no narrowphase, 2 = no pair
                                          // loop, 4 = no queue appends, 8 = no GJK phase, 16 = no FK replay, 32 = no cores, 64 = no pre-check,
                                          // 128 = k_narrow accumulates per-phase cycle counts
};
#ifdef NBK_ABLATE_BUILD
#define NBK_DBG(m) ((m).dbg)
#else
#define NBK_DBG(m) 0
#endif

}  // namespace nbk

namespace nbk {
// optional q source of k_broad / k_narrow: configuration b is sample `map[b] & 0xffffffff` of edge `map[b] >> 32`,
// q = (1-t)*start + t*goal (unfused, like SciPy's degree-1 de Boor), t = i*step for i < n, T_f for i == n
struct EdgeSrc {
    const double* starts;
    const double* goals;
    const double* plan;                  // [E][3] step, T_f, n (as double)
    const unsigned long long* map;       // [capacity] or nullptr = plain q rows
    const unsigned long long* total;     // edge mode: device address of the sample count (known only on the device: the launch
                                         // covers the scratch's capacity, blocks beyond the count exit at once)
    long long b0;                        // first configuration of this tile within the flat batch
    unsigned char* ovf;                  // [blocks of the tile] set by a block whose items did not fit their sub-queue (queues are sized
                                         // for a budget, not for the worst case): k_validity_redo re-decides that block without a queue
};
// configurations this launch really has: the tile size B, or what is left of the device-side sample count
NBK_DEV int64_t effective_batch(const EdgeSrc& es, int64_t B) {
    if (es.total == nullptr) return B;
    const long long left = (long long)(*es.total) - es.b0;
    return left < (long long)B ? (left < 0 ? 0 : (int64_t)left) : B;
}
}  // namespace nbk

namespace nbk {
// per-(descriptor, stream) scratch.  The float32 broadphase tables stay valid while the threshold does not change and the queue
// counters exist twice -- a call uses set (epoch & 1), its narrowphase clears the other set for the next call -- so steady-state
// calls launch two kernels, not three.  The edge path keeps plan / counts / offsets / sample map / mask words here, sized for a
// capacity in samples; `stats` is pinned host memory the device writes the true sample count into (read, never waited for, at the
// start of the NEXT call to grow the capacity).
struct StreamWs {
    hipStream_t stream;
    std::mutex mu;                  // two host threads driving one stream
    void* ws; size_t ws_bytes;
    bool ready; double thr; unsigned epoch;
    bool captured;                  // a call on this stream has been captured into a hipGraph: its nodes reuse this workspace (counter
                                    // set 0, the tables for THEIR threshold) whenever the graph is replayed, behind the host's back, so
                                    // direct calls on this stream never trust `ready` again -- each prepares its tables and clears both
                                    // counter sets itself (one more 5 us launch per call)
    void* ews; size_t ews_bytes;
    long long ecap_edges; unsigned long long ecap_samples;
    unsigned long long* stats;      // [4] pinned + mapped: samples needed by the last finished edge call, edges served by the overflow kernel
    unsigned long long* stats_dev;  // device alias of `stats`
    // tile pipelining of batches of several tiles: odd tiles run on `aux_stream` with the scratch set `aux` (forked from / joined
    // to the caller's stream with events), so the latency-bound narrowphase of one tile overlaps the issue-bound broadphase of the next
    hipStream_t aux_stream;
    hipEvent_t ev_fork, ev_join;
    StreamWs* aux;
};
}  // namespace nbk

struct nbk_model {
    nbk::DevModel d;
    void* blob;
    size_t blob_bytes;
    int device;
    int n_pairs;
    int n_q;
    int n_joints;
    int cls_count[4];          // pairs per kind class; cls_groups (in d) sub-queues serve each
    // per pair, in broadphase order: what the per-call static reach test (k_prepare_f32) needs, so that the host can count the
    // pairs a call can produce items for at ITS threshold and size queues / tiles for those instead of for every pair
    std::vector<double> h_static, h_m0, h_m1;
    std::vector<int> h_cat, h_cls;
    std::vector<int> h_joint_qidx, h_joint_type, h_joint_kind;   // host copies for make_path
    std::vector<double> gjk_margins;   // (mA, mB) of every pair that can reach GJK: the host picks the narrowphase build per call
    bool world_hulls = false;          // a world shape is a hull: k_broad_f32<S, true>
    bool gjk_any_hull = false;         // ... and whether one of them has a hull core (those always take the distance iteration)
    bool parked_ok;           // all robot cores of 64 configurations fit LDS (fused validity, distances, one-wave-per-edge)
    bool lds_broad_ok;        // the LDS broadphase k_broad fits this scene (else only the register broadphases are used)
    bool margins_zero;        // every pair that can reach GJK (no point core, not point/segment x point/segment) has mA = mB = 0:
                              // with threshold 0 its contact threshold tc is exactly 0 and the boolean walk decides it
    // Internal scratch of nbk_validity_batch / nbk_edge_validity_batch: ONE SET PER STREAM (created on a stream's first call), so
    // calls on different streams share nothing mutable and overlap on the device; `mu` only guards the list itself.
    std::mutex mu;
    std::vector<nbk::StreamWs*> wss;
    // nbk_validity_scalar_host: pinned, device-mapped staging for one configuration + its private stream
    std::mutex scalar_mu;
    double* scalar_q;         // [n_q] host-pinned, read by the kernel through its device alias
    unsigned long long* scalar_out;
    double* scalar_q_dev;     // device aliases of the two pinned buffers
    unsigned long long* scalar_out_dev;
    hipStream_t scalar_stream;
};

namespace nbk {

static thread_local char g_err[256] = "";

static int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return NBK_ERR_HIP;
}
#define NBK_HIP(call)                                           \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(e_, #call);       \
    } while (0)

constexpr int WAVE = 64;

NBK_DEV int core_rows(int kind) { return kind == K_POINT ? 3 : ((kind == K_BOX || kind == K_HULL) ? 12 : 6); }

// sample t of flat-batch entry `mp` = (edge << 32 | sample index): i * step for i < n, T_f for the last sample
NBK_DEV double edge_t(const EdgeSrc& es, unsigned long long mp, unsigned& e_out) {
    const unsigned e = (unsigned)(mp >> 32), i = (unsigned)mp;
    const double* pl = es.plan + 3 * (size_t)e;
    e_out = e;
    return ((double)i < pl[2]) ? (double)i * pl[0] : pl[1];
}

// 16-byte store of an output row.  (Non-temporal stores -- __builtin_nontemporal_store -- were tried for the pose / Jacobian streams:
// 12-15x SLOWER on gfx950, k_fk_frames 0.39 -> 6.2 ms per 1e6; plain stores through L2 it is.)
NBK_DEV void store_stream2(double* dst, double x, double y) {
    double2 v; v.x = x; v.y = y;
    *reinterpret_cast<double2*>(dst) = v;
}

// ---- q staging: rows [B][n_q] -> LDS [n_q][64] ---------------------------------------------------
// The block's slab of q is contiguous (64*n_q doubles); it is read with 16-byte loads where the slab is
// full, then each lane picks its own row out of LDS.
NBK_DEV void stage_q(const double* __restrict__ q, int64_t base, int64_t B, int n_q, double* lds_raw, double* lds_q, int lane) {
    const int64_t rows = (B - base) < WAVE ? (B - base) : WAVE;
    const int total = (int)rows * n_q;
    const double* src = q + base * n_q;
    if (rows == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
        const double2* s2 = reinterpret_cast<const double2*>(src);
        double2* d2 = reinterpret_cast<double2*>(lds_raw);
        for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
    } else {
        for (int i = lane; i < total; i += WAVE) lds_raw[i] = src[i];
    }
    __syncthreads();
    if (lane < rows) {
        for (int j = 0; j < n_q; ++j) lds_q[j * WAVE + lane] = lds_raw[lane * n_q + j];
    } else {
        for (int j = 0; j < n_q; ++j) lds_q[j * WAVE + lane] = 0.0;
    }
    __syncthreads();
}

// A configuration with a NaN or infinite joint value is reported as colliding by every validity path (the planner must
// not accept it); `row` points at the lane's first value, `stride` is the distance between consecutive joints.
NBK_DEV bool row_nonfinite(const double* row, int n, int stride) {
    bool bad = false;
    for (int j = 0; j < n; ++j) bad = bad || !(__builtin_fabs(row[j * stride]) <= 1.7976931348623157e308);
    return bad;
}

// child frame of joint k (robots/helpers.py:43-55 restated with the host-made M0/M1/M2):
//   L = M0 - cos(q) M1 + sin(q) M2 (each element fma(s, M2, fma(-c, M1, M0))),  tl = fma(q, slide, trans),  out = parent * (L, tl).
// Joints whose axis is a coordinate axis of the joint frame (every joint of the usual URDFs) have known zeros in the tables:
// column KZ of M1 and M2, and the two other columns of M0 (the host stores exact +0 there).  joint_kind[k] = KZ names that case and the
// same formula is evaluated with literal zeros -- bit-identical (fma(x, y, +0.0) is what the table would give), but every
// instruction now reads ONE scalar constant instead of two or three, so the v_mov's that fed the extra constants (40 % of the
// VALU stream of k_fk) are gone and column KZ of L stays in scalar registers.  Prismatic joints: L = M0, all scalar.
enum { JK_GENERIC = 3, JK_PRISMATIC = 4 };

template <int KZ>
NBK_DEV void joint_apply_axis(const double* M, const double* toff, const Xf& a, double qk, double s, double c, Xf& o) {
    constexpr int U = (KZ + 1) % 3, V = (KZ + 2) % 3;
    double Lu[3], Lv[3], tl[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Lu[r] = NBK_FMA(s, M[18 + 3 * r + U], NBK_FMA(-c, M[9 + 3 * r + U], 0.0));
        Lv[r] = NBK_FMA(s, M[18 + 3 * r + V], NBK_FMA(-c, M[9 + 3 * r + V], 0.0));
        tl[r] = NBK_FMA(qk, 0.0, toff[r]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a0 = a.R[3 * i], a1 = a.R[3 * i + 1], a2 = a.R[3 * i + 2];
        o.R[3 * i + U] = NBK_FMA(a2, Lu[2], NBK_FMA(a1, Lu[1], a0 * Lu[0]));
        o.R[3 * i + V] = NBK_FMA(a2, Lv[2], NBK_FMA(a1, Lv[1], a0 * Lv[0]));
        o.R[3 * i + KZ] = NBK_FMA(a2, M[6 + KZ], NBK_FMA(a1, M[3 + KZ], a0 * M[KZ]));
        o.t[i] = NBK_FMA(a2, tl[2], NBK_FMA(a1, tl[1], NBK_FMA(a0, tl[0], a.t[i])));
    }
}

NBK_DEV void joint_apply(const DevModel& m, int k, const Xf& parent, double qk, Xf& out) {
    const double* M = m.joint_rot + 27 * k;
    const double* toff = m.joint_trans + 3 * k;
    const double* sl = m.joint_slide + 3 * k;
    const int kind = m.joint_kind[k];                  // wave-uniform: scalar branches
    if (kind == JK_PRISMATIC) {
        double tl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) tl[i] = NBK_FMA(qk, sl[i], toff[i]);
        xf_mul(parent, M, tl, out);                    // L = M0 (s = c = 0 and M1 = M2 = 0 in the general formula)
        return;
    }
    double s = 0.0, c = 0.0;
    nbk_sincos(qk, s, c);
    if (kind == 0) { joint_apply_axis<0>(M, toff, parent, qk, s, c, out); return; }
    if (kind == 1) { joint_apply_axis<1>(M, toff, parent, qk, s, c, out); return; }
    if (kind == 2) { joint_apply_axis<2>(M, toff, parent, qk, s, c, out); return; }
    double L[9], tl[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) L[e] = NBK_FMA(s, M[18 + e], NBK_FMA(-c, M[9 + e], M[e]));
#pragma unroll
    for (int i = 0; i < 3; ++i) tl[i] = NBK_FMA(qk, sl[i], toff[i]);
    xf_mul(parent, L, tl, out);
}

// joint_apply for the kernel that unrolls a path of at most 8 joints at compile time (k_jacobian_reg): the joint's kind comes
// from the launch arguments and its constants from ONE 144-byte block (joint_pk) that the caller loads a joint ahead, instead of a
// kind load -> branch -> table loads -> wait chain per joint.  Same operations in the same order as joint_apply_axis: bit-identical.
struct alignas(16) JPkD { double v[22]; };       // [18..20]: the joint axis (k_jacobian_reg), [21] pad
template <int KZ>
NBK_DEV void joint_apply_axis_pk(const JPkD& jp, const Xf& a, double qk, double s, double c, Xf& o) {
    constexpr int U = (KZ + 1) % 3, V = (KZ + 2) % 3;
    double Lu[3], Lv[3], tl[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Lu[r] = NBK_FMA(s, jp.v[2 * r], NBK_FMA(-c, jp.v[6 + 2 * r], 0.0));
        Lv[r] = NBK_FMA(s, jp.v[2 * r + 1], NBK_FMA(-c, jp.v[6 + 2 * r + 1], 0.0));
        tl[r] = NBK_FMA(qk, 0.0, jp.v[15 + r]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a0 = a.R[3 * i], a1 = a.R[3 * i + 1], a2 = a.R[3 * i + 2];
        o.R[3 * i + U] = NBK_FMA(a2, Lu[2], NBK_FMA(a1, Lu[1], a0 * Lu[0]));
        o.R[3 * i + V] = NBK_FMA(a2, Lv[2], NBK_FMA(a1, Lv[1], a0 * Lv[0]));
        o.R[3 * i + KZ] = NBK_FMA(a2, jp.v[14], NBK_FMA(a1, jp.v[13], a0 * jp.v[12]));
        o.t[i] = NBK_FMA(a2, tl[2], NBK_FMA(a1, tl[1], NBK_FMA(a0, tl[0], a.t[i])));
    }
}
NBK_DEV void joint_apply_pk(const DevModel& m, int k, int kind, const JPkD& jp, const Xf& parent, double qk, Xf& out) {
    if (kind <= 2) {
        double s = 0.0, c = 0.0;
        nbk_sincos(qk, s, c);
        if (kind == 0) joint_apply_axis_pk<0>(jp, parent, qk, s, c, out);
        else if (kind == 1) joint_apply_axis_pk<1>(jp, parent, qk, s, c, out);
        else joint_apply_axis_pk<2>(jp, parent, qk, s, c, out);
        return;
    }
    joint_apply(m, k, parent, qk, out);                 // general revolute / prismatic: the table-driven form
}

// ---- FK of one frame -----------------------------------------------------------------------------
// col / revolute / covered restate joint_qidx / joint_type along the path, so that kernels find them in scalar registers
struct PathArg { int len; int idx[NBK_MAX_JOINTS]; double local[12]; int col[NBK_MAX_JOINTS]; unsigned revolute, covered; unsigned char kind[NBK_MAX_JOINTS]; };

// LDS: the raw q slab (64*n_q doubles) and the output rows (64 * 17 doubles) share one region: every q
// value is in a register before the first pose element is written, so 8.7 KB per wave is all it takes and
// 4+ waves per SIMD stay resident to cover the HBM latency (the kernel is bandwidth-bound: 184 B per pose).
// DQ (n_q <= 8): every lane loads its own q row straight into registers -- eight-byte loads at a 8 n_q-byte stride, the wave's
// n_q instructions hit the same lines -- and a joint picks its value with a scalar switch; no LDS pass and no barrier before the sweep.
NBK_DEV double pick8(const double (&v)[8], int i) {
    switch (i) { case 0: return v[0]; case 1: return v[1]; case 2: return v[2]; case 3: return v[3];
                 case 4: return v[4]; case 5: return v[5]; case 6: return v[6]; default: return v[7]; }
}
template <bool DQ>
__global__ __launch_bounds__(64) void k_fk(DevModel m, PathArg path, const double* __restrict__ q, int64_t B,
                                            const double* __restrict__ local_pose, double* __restrict__ T_out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nq = m.n_q;
    const int64_t rows = (B - base) < WAVE ? (B - base) : WAVE;
    double qv[8];
    if constexpr (DQ) {
        const int64_t br = (base + lane) < B ? (base + lane) : (B - 1);
        const double* src = q + br * nq;
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = j < nq ? src[j] : 0.0;
    } else {
        const int total = (int)rows * nq;
        const double* src = q + base * nq;
        if (rows == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(lds);
            for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += WAVE) lds[i] = src[i];
            for (int i = total + lane; i < WAVE * nq; i += WAVE) lds[i] = 0.0;
        }
        __syncthreads();
    }
    Xf T;
    xf_from12(m.base_pose, T);
    const double* myq = lds + lane * nq;
    {
        // two joints per trip (T -> U -> T): no register copies of the 3x4 frame across the loop back-edge
        int i = 0;
        for (; i + 1 < path.len; i += 2) {
            const int k0 = path.idx[i], k1 = path.idx[i + 1];
            const double q0 = DQ ? pick8(qv, m.joint_qidx[k0]) : myq[m.joint_qidx[k0]];
            const double q1 = DQ ? pick8(qv, m.joint_qidx[k1]) : myq[m.joint_qidx[k1]];
            Xf U;
            joint_apply(m, k0, T, q0, U);
            joint_apply(m, k1, U, q1, T);
        }
        if (i < path.len) {
            const int k = path.idx[i];
            Xf U;
            joint_apply(m, k, T, DQ ? pick8(qv, m.joint_qidx[k]) : myq[m.joint_qidx[k]], U);
            T = U;
        }
    }
    Xf loc, E;
    xf_from12(path.local, loc);
    xf_mul(T, loc.R, loc.t, E);
    const int64_t b = base + lane;
    if (local_pose != nullptr && b < B) {
        const double* lp = local_pose + 16 * b;
        Xf P, E2;
        xf_from12(lp, P);
        xf_mul(E, P.R, P.t, E2);
        E = E2;
    }
    if constexpr (!DQ) __syncthreads();   // all q reads are done: the region is reused for the transposed poses
    // transpose through LDS (row stride 17 doubles: conflict-free ds_write_b64)
    double* row = lds + lane * 17;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        row[4 * i] = E.R[3 * i]; row[4 * i + 1] = E.R[3 * i + 1]; row[4 * i + 2] = E.R[3 * i + 2];
        row[4 * i + 3] = E.t[i];
    }
    row[12] = 0.0; row[13] = 0.0; row[14] = 0.0; row[15] = 1.0;
    __syncthreads();
    double2* dst = reinterpret_cast<double2*>(T_out + base * 16);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int g = lane + WAVE * kk;   // double2 index inside the block's 1024-double slab
        const int r = g >> 3, c2 = (g & 7) * 2;
        if (r < rows) store_stream2(reinterpret_cast<double*>(dst + g), lds[r * 17 + c2], lds[r * 17 + c2 + 1]);
    }
}

// ---- geometric Jacobian of one frame ---------------------------------------------------------------
// Two sweeps instead of parking every joint frame in LDS: the first finds the end position, the second
// recomputes the joint frames and writes the columns [w x (p_end - p_i); w] straight into the output rows.
// LDS: raw q slab (64*n_q) | output rows 64 * stride  (25 KB for 7 DoF -> 6 waves per CU instead of 3).
__global__ __launch_bounds__(64) void k_jacobian(DevModel m, PathArg path, const double* __restrict__ q, int64_t B,
                                                  int mode, const double* __restrict__ pose, double* __restrict__ J_out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nq = m.n_q;
    const int64_t rows = (B - base) < WAVE ? (B - base) : WAVE;
    double* lds_o = lds + WAVE * nq;
    {
        const int total = (int)rows * nq;
        const double* src = q + base * nq;
        if (rows == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(lds);
            for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += WAVE) lds[i] = src[i];
            for (int i = total + lane; i < WAVE * nq; i += WAVE) lds[i] = 0.0;
        }
        __syncthreads();
    }
    const double* myq = lds + lane * nq;
    // ---- sweep 1: end position -------------------------------------------------------------------------
    Xf T;
    xf_from12(m.base_pose, T);
    for (int i = 0; i < path.len; ++i) {
        const int k = path.idx[i];
        Xf nxt;
        joint_apply(m, k, T, myq[m.joint_qidx[k]], nxt);
        T = nxt;
    }
    Xf loc, E;
    xf_from12(path.local, loc);
    xf_mul(T, loc.R, loc.t, E);
    const int64_t b = base + lane;
    double pend[3] = {E.t[0], E.t[1], E.t[2]};
    if (mode == 1 && b < B) {
        Xf P;
        xf_from12(pose + 16 * b, P);
        xf_mul_pos(E, P.t, pend);
    } else if (mode == 2 && b < B) {
        pend[0] = pose[16 * b + 3]; pend[1] = pose[16 * b + 7]; pend[2] = pose[16 * b + 11];
    }
    // ---- sweep 2: columns ------------------------------------------------------------------------------
    const int ncol = 6 * nq;
    const int stride = ncol + 1 + ((ncol + 1) & 1 ? 0 : 1);   // odd row stride: conflict-free
    double* row = lds_o + lane * stride;
    for (int c = 0; c < ncol; ++c) row[c] = 0.0;
    xf_from12(m.base_pose, T);
    for (int i = 0; i < path.len; ++i) {
        const int k = path.idx[i];
        const int col = m.joint_qidx[k];
        Xf nxt;
        joint_apply(m, k, T, myq[col], nxt);
        T = nxt;
        const double* a = m.joint_axis + 3 * k;
        double w[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) w[r] = NBK_FMA(T.R[3 * r + 2], a[2], NBK_FMA(T.R[3 * r + 1], a[1], T.R[3 * r] * a[0]));
        if (m.joint_type[k] == NBK_REVOLUTE) {
            double d[3], v[3];
            sub3(pend, T.t, d);
            cross3(w, d, v);
            row[0 * nq + col] = v[0]; row[1 * nq + col] = v[1]; row[2 * nq + col] = v[2];
            row[3 * nq + col] = w[0]; row[4 * nq + col] = w[1]; row[5 * nq + col] = w[2];
        } else {
            row[0 * nq + col] = w[0]; row[1 * nq + col] = w[1]; row[2 * nq + col] = w[2];
        }
    }
    __syncthreads();
    double* dst = J_out + base * ncol;
    const int total = (int)rows * ncol;
    if ((ncol & 1) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        // 16-byte stores; (row, column) of the running element are advanced without divisions
        int g = 2 * lane;
        int r = g / ncol, c = g - r * ncol;
        for (; g < total; g += 2 * WAVE) {
            double2 v;
            v.x = lds_o[r * stride + c];
            v.y = lds_o[r * stride + c + 1];
            *reinterpret_cast<double2*>(dst + g) = v;
            c += 2 * WAVE;
            while (c >= ncol) { c -= ncol; ++r; }
        }
    } else {
        int g = lane;
        int r = g / ncol, c = g - r * ncol;
        for (; g < total; g += WAVE) {
            dst[g] = lds_o[r * stride + c];
            c += WAVE;
            while (c >= ncol) { c -= ncol; ++r; }
        }
    }
}


// ---- geometric Jacobian, paths of at most NJ joints: one sweep, joint axes and origins kept in registers ---------------
// k_jacobian above sweeps the path twice and stages all 6*n_q output rows in LDS (25 KB per wave on a 7-DoF arm: 1.3
// waves per SIMD resident).  Here the world axis w_i and origin o_i of every joint stay in registers (6 NJ doubles), the
// columns are assembled after the single sweep, and the output goes through LDS 16 configurations at a time (see the
// output stage below): half the arithmetic, 9 KB of LDS per wave.  Same formulas, same bits.
constexpr int JAC_ROWS = 32;
// One-wave workgroups exchange data through LDS without s_barrier: LDS executes a wave's instructions in order, so
// a read issued after a write sees it.  Only the compiler has to keep the order, and outstanding global stores are
// NOT waited for (what __syncthreads would do) -- the stores of one group overlap the assembly of the next.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
template <int NJ>
__global__ __launch_bounds__(64) void k_jacobian_reg(DevModel m, PathArg path, const double* __restrict__ q, int64_t B,
                                                      int mode, const double* __restrict__ pose, double* __restrict__ J_out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nq = m.n_q;
    const int64_t rows = (B - base) < WAVE ? (B - base) : WAVE;
    // (mode bit 8: stage q through LDS as the general kernel does -- the A/B switch NBK_FK_LDS_Q; otherwise every lane loads its own
    // values straight from its row: the wave's loads hit the same lines, and there is no LDS pass or barrier before the sweep)
    const bool stage = (mode & 256) != 0;
    mode &= 255;
    if (stage) {
        const int total = (int)rows * nq;
        const double* src = q + base * nq;
        if (rows == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(lds);
            for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += WAVE) lds[i] = src[i];
            for (int i = total + lane; i < WAVE * nq; i += WAVE) lds[i] = 0.0;
        }
        __syncthreads();
    }
    const double* myq = stage ? lds + lane * nq : q + ((base + lane) < B ? (base + lane) : (B - 1)) * nq;
    double Wx[NJ][3], Ox[NJ][3];
    Xf T;
    xf_from12(m.base_pose, T);
    // every q value of the path up front, every joint's constants one joint ahead (joint_apply_pk)
    double qv[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) qv[i] = myq[path.col[i]];
    JPkD jcur = *reinterpret_cast<const JPkD*>(m.joint_pk + 22 * path.idx[0]);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const JPkD jnxt = *reinterpret_cast<const JPkD*>(m.joint_pk + 22 * path.idx[i + 1 < NJ ? i + 1 : NJ - 1]);
#pragma unroll
        for (int r = 0; r < 3; ++r) { Wx[i][r] = 0.0; Ox[i][r] = 0.0; }
        if (i < path.len) {
            const int k = path.idx[i];
            Xf nxt;
            joint_apply_pk(m, k, (int)path.kind[i], jcur, T, qv[i], nxt);
            T = nxt;
            const double* a = jcur.v + 18;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                Wx[i][r] = NBK_FMA(T.R[3 * r + 2], a[2], NBK_FMA(T.R[3 * r + 1], a[1], T.R[3 * r] * a[0]));
                Ox[i][r] = T.t[r];
            }
        }
        jcur = jnxt;
    }
    Xf loc, E;
    xf_from12(path.local, loc);
    xf_mul(T, loc.R, loc.t, E);
    const int64_t b = base + lane;
    double pend[3] = {E.t[0], E.t[1], E.t[2]};
    if (mode == 1 && b < B) {
        Xf P;
        xf_from12(pose + 16 * b, P);
        xf_mul_pos(E, P.t, pend);
    } else if (mode == 2 && b < B) {
        pend[0] = pose[16 * b + 3]; pend[1] = pose[16 * b + 7]; pend[2] = pose[16 * b + 11];
    }
    // Columns: v_i = w_i x (p_end - o_i) replaces o_i in its registers (a prismatic joint's linear column is w_i itself).
    const unsigned covered = path.covered, revolute = path.revolute;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        if (i < path.len) {
            if ((revolute >> i) & 1u) {
                double d[3], v[3];
                sub3(pend, Ox[i], d);
                cross3(Wx[i], d, v);
                Ox[i][0] = v[0]; Ox[i][1] = v[1]; Ox[i][2] = v[2];
            } else {
                Ox[i][0] = Wx[i][0]; Ox[i][1] = Wx[i][1]; Ox[i][2] = Wx[i][2];
            }
        }
    }
    // Output: JAC_ROWS configurations at a time.  The lanes of a group put their full rows (6 n_q doubles) into LDS (the q
    // staging area is free by now and is reused), then the whole wave streams the group's JAC_ROWS * 6 n_q contiguous doubles
    // out with 16-byte stores: every store instruction covers whole, contiguous cache lines.
    const int ncol = 6 * nq;
    const int stride = ncol | 1;                              // odd row stride: conflict-free
    double* lds_o = lds;
    double* row = lds_o + (lane & (JAC_ROWS - 1)) * stride;
    double* dst = J_out + base * ncol;
    const bool wide = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    const unsigned all_cols = nq >= 32 ? 0xffffffffu : ((1u << nq) - 1u);
    // the streaming loop advances (row, column) of its running element by a fixed step, without divisions
    const int per = wide ? 2 * WAVE : WAVE;
    const int step_r = per / ncol, step_c = per - step_r * ncol;
    wave_lds_sync();
    for (int grp = 0; grp < WAVE / JAC_ROWS; ++grp) {
        if (grp * JAC_ROWS >= rows) break;
        if ((lane / JAC_ROWS) == grp) {
            if (covered != all_cols) {
                for (int c = 0; c < nq; ++c)
                    if (!((covered >> c) & 1u))
                        for (int r = 0; r < 6; ++r) row[r * nq + c] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < NJ; ++i) {
                if (i < path.len) {
                    double* rc = row + path.col[i];
                    const bool rev = (revolute >> i) & 1u;
                    rc[0] = Ox[i][0]; rc += nq;
                    rc[0] = Ox[i][1]; rc += nq;
                    rc[0] = Ox[i][2]; rc += nq;
                    rc[0] = rev ? Wx[i][0] : 0.0; rc += nq;
                    rc[0] = rev ? Wx[i][1] : 0.0; rc += nq;
                    rc[0] = rev ? Wx[i][2] : 0.0;
                }
            }
        }
        wave_lds_sync();
        const int nr = (int)rows - grp * JAC_ROWS < JAC_ROWS ? (int)rows - grp * JAC_ROWS : JAC_ROWS;
        const int total = nr * ncol;
        double* d = dst + (size_t)grp * JAC_ROWS * ncol;
        int g = wide ? 2 * lane : lane;
        int r = g / ncol, c = g - r * ncol;
        int o = r * stride + c;
        if (wide) {
            for (; g < total; g += 2 * WAVE) {
                store_stream2(d + g, lds_o[o], lds_o[o + 1]);
                c += step_c; o += step_r * stride + step_c;
                if (c >= ncol) { c -= ncol; o += stride - ncol; }
            }
        } else {
            for (; g < total; g += WAVE) {
                d[g] = lds_o[o];
                c += step_c; o += step_r * stride + step_c;
                if (c >= ncol) { c -= ncol; o += stride - ncol; }
            }
        }
        wave_lds_sync();
    }
}


// ---- FK of many frames in one sweep (all link poses of a configuration) ---------------------------------------------
// One tree sweep with the descriptor's load/save plan; after joint k the poses of the requested frames that hang off it
// (fs_begin ranges, frames sorted by joint) are E = T_k * local_f, transposed through LDS (17-double rows, as k_fk) and
// written as whole 128-byte lines to T_out[b][f_out][16].  56 + 128 n_frames bytes per configuration: HBM-bound.
// LDS: q rows [n_q][64] | saved frames [12 slots][64] | transpose 64 x 17.
__global__ __launch_bounds__(64) void k_fk_frames(DevModel m, const double* __restrict__ q, int64_t B, int n_frames,
                                                   const int* __restrict__ fs_begin, const int* __restrict__ fs_out,
                                                   const double* __restrict__ fs_local, double* __restrict__ T_out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    double* lds_q = lds;
    double* lds_fr = lds_q + WAVE * m.n_q;
    double* lds_t = lds_fr + WAVE * 12 * m.frame_slots;
    stage_q(q, base, B, m.n_q, lds_t, lds_q, lane);         // the transpose area doubles as the raw q slab (n_q <= 17 or larger: sized by the host)
    const int64_t rows = (B - base) < WAVE ? (B - base) : WAVE;
    Xf bpose;
    xf_from12(m.base_pose, bpose);
    Xf T = bpose;
    for (int k = -1; k < m.n_joints; ++k) {
        if (k >= 0) {
            const int ld = m.joint_load[k];
            Xf P;
            if (ld == -2) P = T;
            else if (ld == -1) P = bpose;
            else {
#pragma unroll
                for (int e = 0; e < 9; ++e) P.R[e] = lds_fr[(ld * 12 + e) * WAVE + lane];
#pragma unroll
                for (int e = 0; e < 3; ++e) P.t[e] = lds_fr[(ld * 12 + 9 + e) * WAVE + lane];
            }
            joint_apply(m, k, P, lds_q[m.joint_qidx[k] * WAVE + lane], T);
            const int sv = m.joint_save[k];
            if (sv >= 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) lds_fr[(sv * 12 + e) * WAVE + lane] = T.R[e];
#pragma unroll
                for (int e = 0; e < 3; ++e) lds_fr[(sv * 12 + 9 + e) * WAVE + lane] = T.t[e];
            }
        }
        const int f0 = fs_begin[k + 1], f1 = fs_begin[k + 2];
        for (int f = f0; f < f1; ++f) {
            Xf loc, E;
            xf_from12(fs_local + 12 * f, loc);
            if (k < 0) xf_mul(bpose, loc.R, loc.t, E); else xf_mul(T, loc.R, loc.t, E);
            wave_lds_sync();                                   // the previous frame's rows have been read (stores stay in flight)
            double* row = lds_t + lane * 17;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                row[4 * i] = E.R[3 * i]; row[4 * i + 1] = E.R[3 * i + 1]; row[4 * i + 2] = E.R[3 * i + 2];
                row[4 * i + 3] = E.t[i];
            }
            row[12] = 0.0; row[13] = 0.0; row[14] = 0.0; row[15] = 1.0;
            wave_lds_sync();
            const int fo = fs_out[f];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int g = lane + WAVE * kk;                // double2 index inside this frame's 64 x 16 slab
                const int r = g >> 3, c2 = (g & 7) * 2;
                if (r < rows) store_stream2(T_out + ((size_t)(base + r) * n_frames + fo) * 16 + c2, lds_t[r * 17 + c2], lds_t[r * 17 + c2 + 1]);
            }
        }
    }
}

// ---- exact k nearest neighbours among the points inserted before -----------------------------
// What an insert-then-query loop over faiss.IndexFlatL2 (numbotics/math/geometry/nearest_neighbors.py:6-85, used by
// PlanningGraph.k_nearest, planning/sampling_based/graph.py:165-178) returns for point i: the k points of 0..i with the
// smallest float32 squared L2 distance.  One wave per query; lane l keeps the l-th smallest (distance, index) key seen
// so far (k <= 64, keys are unique so ties order by index); every chunk of 64 candidates is bitonic-sorted across the
// lanes and merged -- skipped outright when no candidate beats the current k-th key.
// distance = sum over the dimensions, in order, of (x - y)^2 with separate float32 roundings (no fma).
NBK_DEV unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, mask, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}
NBK_DEV unsigned long long shfl_u64(unsigned long long v, int src) {
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src, 64);
    const unsigned hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// ascending bitonic merge of a bitonic sequence spread over the 64 lanes
NBK_DEV unsigned long long bitonic_merge64(unsigned long long v, int lane) {
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
        const unsigned long long o = shfl_xor_u64(v, j);
        const bool lower = (lane & j) == 0;
        v = (lower == (v < o)) ? v : o;
    }
    return v;
}
NBK_DEV unsigned long long bitonic_sort64(unsigned long long v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const unsigned long long o = shfl_xor_u64(v, j);
            const bool up = (lane & k) == 0 || k == 64;          // final pass: whole wave ascending
            const bool lower = (lane & j) == 0;
            const bool take_min = lower == up;
            v = (take_min == (v < o)) ? v : o;
        }
    }
    return v;
}

__global__ __launch_bounds__(64) void k_knn_prefix(const float* __restrict__ pts, int N, int dim, int k, int32_t* __restrict__ out) {
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const unsigned long long INF_KEY = ~0ull;
    unsigned long long best = INF_KEY;                            // lane l: l-th smallest key so far
    const float* x = pts + (size_t)i * dim;
    for (int j0 = 0; j0 <= i; j0 += WAVE) {
        const int j = j0 + lane;
        unsigned long long key = INF_KEY;
        if (j <= i) {
            const float* y = pts + (size_t)j * dim;
            float d = 0.0f;
            for (int c = 0; c < dim; ++c) { const float t = x[c] - y[c]; const float t2 = t * t; d = d + t2; }
            key = ((unsigned long long)__builtin_bit_cast(unsigned, d) << 32) | (unsigned)j;
        }
        const unsigned long long kth = shfl_u64(best, k - 1);
        if (__builtin_amdgcn_ballot_w64(key < kth) == 0ull) continue;
        key = bitonic_sort64(key, lane);
        // the 64 smallest of best (ascending) and key (ascending): lane-wise min against the reversed other list
        const unsigned long long rev = shfl_u64(key, 63 - lane);
        best = bitonic_merge64(best < rev ? best : rev, lane);
    }
    if (lane < k) out[(size_t)i * k + lane] = best == INF_KEY ? -1 : (int32_t)(unsigned)(best & 0xFFFFFFFFull);
}

// ---- batched Levenberg-Marquardt inverse kinematics (Arm.inverse_kinematics, robots/arm.py:464-552) ----------
// One problem per lane, the whole iteration in one launch:
//   q <- q + J^T (J J^T + lambda I)^-1 diff,  optional clip to the joint limits,  FK,  diff = [p* - p ; vee(0.5 (R - R^T))]
//   with R = R* R_ee^T (math/spatial.py:207-212);  lambda *= 1.2 and failures += 1 when |diff| grew, else
//   lambda *= 0.5 and failures = 0;  an element runs while |diff| > tol and failures < max_failures.
// One sweep of the path gives both the pose (for this iteration's diff) and the Jacobian (for the next step).  The
// 6x6 system is symmetric positive definite (lambda > 0): an unpivoted Cholesky in registers, where the reference
// calls LAPACK's LU (np.linalg.solve) -- equal to rounding, not bit for bit; a non-positive pivot stops the element.
// LDS rows of 64 doubles: [n_q q][6 n_q J][6 path_len joint axes and origins].
struct IkArg { double tol; int max_iter; int max_failures; int use_limits; double lo[NBK_MAX_DOF]; double hi[NBK_MAX_DOF]; };

NBK_DEV void ik_sweep(const DevModel& m, const PathArg& path, const double* lds_q, double* lds_J, double* lds_jz, int lane, Xf& E) {
    Xf T;
    xf_from12(m.base_pose, T);
    for (int i = 0; i < path.len; ++i) {
        const int k = path.idx[i];
        Xf nxt;
        joint_apply(m, k, T, lds_q[path.col[i] * WAVE + lane], nxt);
        T = nxt;
        const double* a = m.joint_axis + 3 * k;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            lds_jz[(6 * i + r) * WAVE + lane] = NBK_FMA(T.R[3 * r + 2], a[2], NBK_FMA(T.R[3 * r + 1], a[1], T.R[3 * r] * a[0]));
            lds_jz[(6 * i + 3 + r) * WAVE + lane] = T.t[r];
        }
    }
    Xf loc;
    xf_from12(path.local, loc);
    xf_mul(T, loc.R, loc.t, E);
    const int nq = m.n_q;
    for (int i = 0; i < path.len; ++i) {
        const int col = path.col[i];
        const double w[3] = {lds_jz[(6 * i) * WAVE + lane], lds_jz[(6 * i + 1) * WAVE + lane], lds_jz[(6 * i + 2) * WAVE + lane]};
        if ((path.revolute >> i) & 1u) {
            const double o[3] = {lds_jz[(6 * i + 3) * WAVE + lane], lds_jz[(6 * i + 4) * WAVE + lane], lds_jz[(6 * i + 5) * WAVE + lane]};
            double d[3], v[3];
            sub3(E.t, o, d);
            cross3(w, d, v);
#pragma unroll
            for (int r = 0; r < 3; ++r) { lds_J[(r * nq + col) * WAVE + lane] = v[r]; lds_J[((3 + r) * nq + col) * WAVE + lane] = w[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) { lds_J[(r * nq + col) * WAVE + lane] = w[r]; lds_J[((3 + r) * nq + col) * WAVE + lane] = 0.0; }
        }
    }
}

NBK_DEV double ik_diff(const Xf& P, const Xf& E, double* d) {
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = P.t[i] - E.t[i];
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = NBK_FMA(P.R[3 * i + 2], E.R[3 * j + 2], NBK_FMA(P.R[3 * i + 1], E.R[3 * j + 1], P.R[3 * i] * E.R[3 * j]));
    d[3] = 0.5 * (R[2][1] - R[1][2]);
    d[4] = 0.5 * (R[0][2] - R[2][0]);
    d[5] = 0.5 * (R[1][0] - R[0][1]);
    double s = d[0] * d[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) s = NBK_FMA(d[i], d[i], s);
    return nbk_sqrt(s);
}

// x = (J J^T + lambda I)^-1 d by Cholesky; false when a pivot is not positive
NBK_DEV bool ik_solve(const double* lds_J, int nq, int lane, double lambda, const double* d, double* x) {
    double A[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) A[r][c] = 0.0;
    for (int j = 0; j < nq; ++j) {
        double cj[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) cj[r] = lds_J[(r * nq + j) * WAVE + lane];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) A[r][c] = NBK_FMA(cj[r], cj[c], A[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) A[r][r] = A[r][r] + lambda;
    double L[6][6];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sum = A[j][i];
#pragma unroll
            for (int k = 0; k < j; ++k) sum = NBK_FMA(-L[i][k], L[j][k], sum);
            if (i == j) { if (!(sum > 0.0)) ok = false; L[i][i] = nbk_sqrt(sum); }
            else L[i][j] = sum / L[j][j];
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sum = d[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum = NBK_FMA(-L[i][k], y[k], sum);
        y[i] = sum / L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sum = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) sum = NBK_FMA(-L[k][i], x[k], sum);
        x[i] = sum / L[i][i];
    }
    return ok;
}

__global__ __launch_bounds__(64) void k_ik(DevModel m, PathArg path, IkArg arg, const double* __restrict__ pose,
                                            const double* __restrict__ q0, int64_t B, double* __restrict__ q_out,
                                            uint8_t* __restrict__ success, double* __restrict__ diff_norm, int32_t* __restrict__ iters) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int nq = m.n_q;
    const int64_t b = (int64_t)blockIdx.x * WAVE + lane;
    const bool active = b < B;
    double* lds_q = lds;
    double* lds_J = lds_q + WAVE * nq;
    double* lds_jz = lds_J + WAVE * 6 * nq;
    for (int j = 0; j < nq; ++j) lds_q[j * WAVE + lane] = active ? q0[b * nq + j] : 0.0;
    for (int r = 0; r < 6 * nq; ++r) lds_J[r * WAVE + lane] = 0.0;
    Xf P;
    xf_from12(m.base_pose, P);
    if (active) xf_from12(pose + 16 * b, P);
    Xf E;
    ik_sweep(m, path, lds_q, lds_J, lds_jz, lane, E);
    double d[6];
    double nrm = ik_diff(P, E, d);
    double lambda = 1e-1;
    int fail = 0, used = 0;
    bool running = active && (nrm > arg.tol) && (fail < arg.max_failures);
    for (int it = 0; it < arg.max_iter; ++it) {
        if (__builtin_amdgcn_ballot_w64(running) == 0ull) break;
        if (running) {
            double x[6];
            const bool ok = ik_solve(lds_J, nq, lane, lambda, d, x);
            if (!ok) {
                fail = arg.max_failures;           // singular damped system: the element stops, unsolved
            } else {
                for (int j = 0; j < nq; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc = NBK_FMA(lds_J[(r * nq + j) * WAVE + lane], x[r], acc);
                    double qj = lds_q[j * WAVE + lane] + acc;
                    if (arg.use_limits) { if (qj < arg.lo[j]) qj = arg.lo[j]; if (qj > arg.hi[j]) qj = arg.hi[j]; }
                    lds_q[j * WAVE + lane] = qj;
                }
                ik_sweep(m, path, lds_q, lds_J, lds_jz, lane, E);
                const double prev = nrm;
                nrm = ik_diff(P, E, d);
                const bool grew = nrm > prev;
                lambda = lambda * (grew ? 1.2 : 0.5);
                fail = grew ? fail + 1 : 0;
                used += 1;
            }
        }
        running = running && (nrm > arg.tol) && (fail < arg.max_failures);
    }
    if (active) {
        for (int j = 0; j < nq; ++j) q_out[b * nq + j] = lds_q[j * WAVE + lane];
        success[b] = (nrm < arg.tol) ? 1 : 0;
        if (diff_norm != nullptr) diff_norm[b] = nrm;
        if (iters != nullptr) iters[b] = used;
    }
}

// ---- collision: sweep the tree, park robot cores in LDS ---------------------------------------------
// LDS rows of 64 doubles: [n_q q rows][shape_rows][frame_slots * 12]
// lds_jz (optional): [J][6] rows -- world axis w_k = R_k a_k and origin o_k of every joint, for Jacobian rows
NBK_DEV void sweep_and_park(const DevModel& m, double* lds_q, double* lds_s, double* lds_fr, int lane, double* lds_jz = nullptr) {
    Xf base;
    xf_from12(m.base_pose, base);
    Xf T = base;
    for (int k = -1; k < m.n_joints; ++k) {
        if (k >= 0) {
            const int ld = m.joint_load[k];
            Xf P;
            if (ld == -2) P = T;
            else if (ld == -1) P = base;
            else {
#pragma unroll
                for (int e = 0; e < 9; ++e) P.R[e] = lds_fr[(ld * 12 + e) * WAVE + lane];
#pragma unroll
                for (int e = 0; e < 3; ++e) P.t[e] = lds_fr[(ld * 12 + 9 + e) * WAVE + lane];
            }
            const double qk = lds_q[m.joint_qidx[k] * WAVE + lane];
            joint_apply(m, k, P, qk, T);
            if (lds_jz != nullptr) {
                const double* a = m.joint_axis + 3 * k;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    lds_jz[(6 * k + r) * WAVE + lane] = NBK_FMA(T.R[3 * r + 2], a[2], NBK_FMA(T.R[3 * r + 1], a[1], T.R[3 * r] * a[0]));
                    lds_jz[(6 * k + 3 + r) * WAVE + lane] = T.t[r];
                }
            }
            const int sv = m.joint_save[k];
            if (sv >= 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) lds_fr[(sv * 12 + e) * WAVE + lane] = T.R[e];
#pragma unroll
                for (int e = 0; e < 3; ++e) lds_fr[(sv * 12 + 9 + e) * WAVE + lane] = T.t[e];
            }
        }
        const int s0 = m.joint_shape_begin[k + 1], s1 = m.joint_shape_begin[k + 2];
        for (int s = s0; s < s1; ++s) {
            const double* loc = m.rs_local + 12 * s;
            double Rl[9], tl[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { Rl[3 * i] = loc[4 * i]; Rl[3 * i + 1] = loc[4 * i + 1]; Rl[3 * i + 2] = loc[4 * i + 2]; tl[i] = loc[4 * i + 3]; }
            double* rows = lds_s + m.rs_row[s] * WAVE + lane;
            double c[3];
            xf_mul_pos(T, tl, c);
            rows[0] = c[0]; rows[WAVE] = c[1]; rows[2 * WAVE] = c[2];
            const int kind = m.rs_kind[s];
            if (kind == K_SEG || kind == K_CYL) {
                double u[3];
                xf_mul_col(T, Rl, 2, u);
                rows[3 * WAVE] = u[0]; rows[4 * WAVE] = u[1]; rows[5 * WAVE] = u[2];
            } else if (kind == K_BOX || kind == K_HULL) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double u[3];
                    xf_mul_col(T, Rl, j, u);
                    rows[(3 + 3 * j) * WAVE] = u[0]; rows[(4 + 3 * j) * WAVE] = u[1]; rows[(5 + 3 * j) * WAVE] = u[2];
                }
            }
        }
    }
}

NBK_DEV void load_rcore(const DevModel& m, const double* lds_s, int s, int lane, Core& o) {
    const double* rows = lds_s + m.rs_row[s] * WAVE + lane;
    const double* cc = m.rs_core + 6 * s;
    o.kind = m.rs_kind[s];
    o.h[0] = cc[0]; o.h[1] = cc[1]; o.h[2] = cc[2]; o.rad = cc[3]; o.margin = cc[4]; o.rho = cc[5];
    o.c[0] = rows[0]; o.c[1] = rows[WAVE]; o.c[2] = rows[2 * WAVE];
    if (o.kind == K_SEG || o.kind == K_CYL) {
        o.ax[2][0] = rows[3 * WAVE]; o.ax[2][1] = rows[4 * WAVE]; o.ax[2][2] = rows[5 * WAVE];
    } else if (o.kind == K_BOX || o.kind == K_HULL) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            o.ax[j][0] = rows[(3 + 3 * j) * WAVE]; o.ax[j][1] = rows[(4 + 3 * j) * WAVE]; o.ax[j][2] = rows[(5 + 3 * j) * WAVE];
        }
    }
}

NBK_DEV void load_wcore(const DevModel& m, int w, Core& o) {
    const double* cc = m.ws_core + 18 * w;
    o.kind = m.ws_kind[w];
    o.c[0] = cc[0]; o.c[1] = cc[1]; o.c[2] = cc[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) { o.ax[j][0] = cc[3 + 3 * j]; o.ax[j][1] = cc[4 + 3 * j]; o.ax[j][2] = cc[5 + 3 * j]; }
    o.h[0] = cc[12]; o.h[1] = cc[13]; o.h[2] = cc[14]; o.rad = cc[15]; o.margin = cc[16]; o.rho = cc[17];
}

NBK_DEV void load_pair(const DevModel& m, const double* lds_s, int p, int lane, Core& A, Core& Bc) {
    const int a = m.pair_a[p], b = m.pair_b[p];
    load_rcore(m, lds_s, a, lane, A);
    if (b < m.n_rshapes) load_rcore(m, lds_s, b, lane, Bc);
    else load_wcore(m, b - m.n_rshapes, Bc);
}

// ---- validity: wavefront broadphase -> compacted narrowphase -------------------------------------------
// Phase A (dense, uniform): every lane runs the bounding-sphere test of every pair on ITS configuration;
//   plane pairs and closed-form pairs that survive are decided on the spot; surviving GJK pairs are appended
//   to a wave-shared LDS queue as (source lane, pair) items with a ballot + prefix count.
// Phase B (dense again): the queue is drained 64 items at a time -- lane i evaluates item i, reading the
//   cores of the SOURCE lane's configuration out of the [component][lane] LDS rows -- and raises the source
//   lane's hit flag.  On the benchmark scene ~2 % of the (configuration, pair) items survive phase A, so the
//   branchy GJK runs on full waves instead of on a few scattered lanes per pair.
constexpr int QUEUE_CAP = 512;                  // items; flushed whenever fewer than 64 slots are left
constexpr int VALIDITY_LDS_EXTRA = QUEUE_CAP * 4 + WAVE * 4;

// core of shape `ref` for the configuration parked in LDS column `col` (ref / col may differ per lane)
NBK_DEV void load_core_any(const DevModel& m, const double* lds_s, int ref, int col, Core& o) {
    if (ref >= 0) {
        const double* rows = lds_s + m.rs_row[ref] * WAVE + col;
        const double* cc = m.rs_core + 6 * ref;
        o.kind = m.rs_kind[ref];
        o.h[0] = cc[0]; o.h[1] = cc[1]; o.h[2] = cc[2]; o.rad = cc[3]; o.margin = cc[4]; o.rho = cc[5];
        o.c[0] = rows[0]; o.c[1] = rows[WAVE]; o.c[2] = rows[2 * WAVE];
        if (o.kind == K_SEG || o.kind == K_CYL) {
            o.ax[2][0] = rows[3 * WAVE]; o.ax[2][1] = rows[4 * WAVE]; o.ax[2][2] = rows[5 * WAVE];
        } else if (o.kind == K_BOX || o.kind == K_HULL) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                o.ax[j][0] = rows[(3 + 3 * j) * WAVE]; o.ax[j][1] = rows[(4 + 3 * j) * WAVE]; o.ax[j][2] = rows[(5 + 3 * j) * WAVE];
            }
        }
    } else {
        load_wcore(m, ~ref, o);
    }
}

NBK_DEV void drain_queue(const DevModel& m, const double* lds_s, const unsigned* queue, int qn, unsigned* lds_hit,
                         int lane, double thr) {
    __syncthreads();            // one wave per workgroup: orders the queue / flag writes before the reads below
    if (NBK_DBG(m) & 1) qn = 0;
    for (int base = 0; base < qn; base += WAVE) {
        const int i = base + lane;
        if (i < qn) {
            const unsigned item = queue[i];
            const int src = (int)(item & 63u);
            const int p = (int)(item >> 6);
            if (lds_hit[src] == 0u) {
                const double* cst = m.vp_cst + 4 * p;
                const double tc = (thr + cst[0]) + cst[1];
                Core A, Bc;
                load_core_any(m, lds_s, m.vp_canon[2 * p], src, A);
                load_core_any(m, lds_s, m.vp_canon[2 * p + 1], src, Bc);
                if (cores_collide_exact(A, Bc, tc)) lds_hit[src] = 1u;
            }
        }
    }
}

// returns this lane's in-collision flag.  lds_x: QUEUE_CAP queue words followed by 64 hit flags.
NBK_DEV bool wave_collides(const DevModel& m, const double* lds_s, unsigned* lds_x, int lane, double thr, bool active) {
    unsigned* queue = lds_x;
    unsigned* lds_hit = lds_x + QUEUE_CAP;
    lds_hit[lane] = 0u;
    bool hit = false;
    int qn = 0;                                       // wave-uniform
    const int np = (NBK_DBG(m) & 2) ? 0 : m.n_pairs;
    for (int p = 0; p < np; ++p) {
        const int* tab = m.vp_tab + 4 * p;
        const int refA = tab[0], refB = tab[1];
        if (p < m.n_plane_pairs) {
            if (active && !hit) {
                Core A, Pl;
                load_core_any(m, lds_s, refA, lane, A);
                load_wcore(m, ~refB, Pl);
                if (plane_collides(A, Pl, thr, m.vp_cst[4 * p + 2])) hit = true;
            }
            continue;
        }
        const double* cst = m.vp_cst + 4 * p;
        const double tc = (thr + cst[0]) + cst[1];
        const double rs = (tc + cst[2]) + cst[3];
        if (!(rs > 0.0)) continue;                    // uniform: the spheres can never be that close
        // centres: robot shapes from this lane's LDS column, world shapes from the scalar table
        const double* ra = lds_s + m.rs_row[refA] * WAVE + lane;
        double dl[3];
        if (refB >= 0) {
            const double* rb = lds_s + m.rs_row[refB] * WAVE + lane;
            dl[0] = ra[0] - rb[0]; dl[1] = ra[WAVE] - rb[WAVE]; dl[2] = ra[2 * WAVE] - rb[2 * WAVE];
        } else {
            const double* wc = m.ws_center + 3 * (~refB);
            dl[0] = ra[0] - wc[0]; dl[1] = ra[WAVE] - wc[1]; dl[2] = ra[2 * WAVE] - wc[2];
        }
        const bool cand = active && !hit && (dot3(dl, dl) < rs * rs);
        if (p < m.n_plane_pairs + m.n_closed_pairs) {
            if (cand) {                               // point / segment cores: decide here
                Core A, Bc;
                load_core_any(m, lds_s, m.vp_canon[2 * p], lane, A);
                load_core_any(m, lds_s, m.vp_canon[2 * p + 1], lane, Bc);
                if (cores_collide_exact(A, Bc, tc)) hit = true;
            }
        } else {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(cand);
            if (bal != 0ull) {
                if (cand) {
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    queue[pos] = ((unsigned)p << 6) | (unsigned)lane;
                }
                qn += __builtin_popcountll(bal);
                if (qn > QUEUE_CAP - WAVE) {
                    if (hit) lds_hit[lane] = 1u;
                    drain_queue(m, lds_s, queue, qn, lds_hit, lane, thr);
                    __syncthreads();
                    qn = 0;
                    hit = hit || (lds_hit[lane] != 0u);
                }
            }
        }
    }
    if (hit) lds_hit[lane] = 1u;
    drain_queue(m, lds_s, queue, qn, lds_hit, lane, thr);
    __syncthreads();
    return lds_hit[lane] != 0u;
}

__global__ __launch_bounds__(64) void k_validity(DevModel m, const double* __restrict__ q, int64_t B, double thr,
                                                  uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    double* lds_q = lds;
    double* lds_s = lds_q + WAVE * m.n_q;
    double* lds_fr = lds_s + WAVE * m.shape_rows;
    unsigned* lds_x = reinterpret_cast<unsigned*>(lds_fr + WAVE * 12 * m.frame_slots);
    // the raw slab is staged in the shape area (free until the sweep starts)
    stage_q(q, base, B, m.n_q, lds_s, lds_q, lane);
    const bool active = (base + lane) < B;
    sweep_and_park(m, lds_q, lds_s, lds_fr, lane);
    const bool bad = row_nonfinite(lds_q + lane, m.n_q, WAVE);
    const bool hit = wave_collides(m, lds_s, lds_x, lane, thr, active) || bad;
    const uint64_t word = __builtin_amdgcn_ballot_w64(hit && active);
    if (mask_bits != nullptr && lane == 0) mask_bits[blockIdx.x] = word;
    if (mask_bytes != nullptr && active) mask_bytes[base + lane] = hit ? 1 : 0;
}

// The queues of the broadphase + narrowphase pipeline are sized for a BUDGET (1 GiB), not for the worst case of every pair of every
// configuration surviving.  A block whose items did not fit marks itself (EdgeSrc::ovf); this kernel then decides the marked blocks
// the queue-less way -- every pair of the 64 configurations, as k_validity does -- and ORs the verdicts into the mask.  Items the
// block did get into the queue were decided by the narrowphase as well: same verdicts twice.  Unmarked blocks exit at once; a
// marked block clears its mark.
__global__ __launch_bounds__(64) void k_validity_redo(DevModel m, EdgeSrc es, const double* __restrict__ q, int64_t B, double thr,
                                                       uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes) {
    extern __shared__ double lds[];
    if (es.ovf[blockIdx.x] == 0) return;
    const int lane = threadIdx.x;
    if (lane == 0) es.ovf[blockIdx.x] = 0;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int64_t Beff = effective_batch(es, B);
    if (base >= Beff) return;
    double* lds_q = lds;
    double* lds_s = lds_q + WAVE * m.n_q;
    double* lds_fr = lds_s + WAVE * m.shape_rows;
    unsigned* lds_x = reinterpret_cast<unsigned*>(lds_fr + WAVE * 12 * m.frame_slots);
    const bool active = (base + lane) < Beff;
    if (es.map != nullptr) {
        const int nq = m.n_q;
        if (active) {
            unsigned e;
            const double t = edge_t(es, es.map[base + lane], e);
            const double omt = 1.0 - t;
            const double* sp = es.starts + (size_t)e * nq;
            const double* gp = es.goals + (size_t)e * nq;
            for (int j = 0; j < nq; ++j) { const double a = omt * sp[j]; const double bb = t * gp[j]; lds_q[j * WAVE + lane] = a + bb; }
        } else {
            for (int j = 0; j < nq; ++j) lds_q[j * WAVE + lane] = 0.0;
        }
        __syncthreads();
    } else {
        stage_q(q, base, Beff, m.n_q, lds_s, lds_q, lane);
    }
    sweep_and_park(m, lds_q, lds_s, lds_fr, lane);
    const bool bad = row_nonfinite(lds_q + lane, m.n_q, WAVE);
    const bool hit = wave_collides(m, lds_s, lds_x, lane, thr, active) || bad;
    const uint64_t word = __builtin_amdgcn_ballot_w64(hit && active);
    if (mask_bits != nullptr && lane == 0 && word != 0ull) atomicOr(reinterpret_cast<unsigned long long*>(mask_bits) + blockIdx.x, (unsigned long long)word);
    if (mask_bytes != nullptr && active && hit) mask_bytes[base + lane] = 1;
}

// ==== two-kernel validity for large batches ===========================================================
// k_broad : one configuration per lane.  Sweeps the tree keeping ONLY the primitive centres (3 LDS rows per
//           shape, so several waves fit a CU), runs the bounding-sphere test of every pair and appends the
//           survivors (configuration, pair) to a GLOBAL queue: per wave one atomicAdd + one coalesced burst.
//           The pair table of a wave lives in VGPRs (lane p holds pair p) and is broadcast with v_readlane,
//           so the pair loop has no scalar-memory latency in it.
// k_narrow: one queue item per lane, items are dense.  Replays the FK of the two primitives of its item
//           from q, runs the exact predicate and ORs the configuration's bit into the mask.
// Both kernels evaluate exactly the predicate of the fused kernel (and of the oracle), so the masks are
// bit-identical; the fused kernel stays for small batches and for the edge kernel.
constexpr int BQ_CAP = 512;             // per-wave LDS staging of queue items before one global append
// k_broad_f32<S>: rows of 64 doubles its q slab / item queue takes.  The queue must hold one whole row of slots (S x 64 items)
// beyond what is pending, so that the "queue nearly full" test -- and with it the inlined flush -- exists once per row / world
// shape instead of once per slot (92 inlined flushes made the kernel 195 KB of code for a 64 KB instruction cache)
__host__ __device__ constexpr int f32_qrows(int nq, int S) { return nq > (S + 2) / 2 ? nq : (S + 2) / 2; }

NBK_DEV double readlane_f64(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// The global queue is sharded into NSUB sub-queues (block b appends to sub-queue b % NSUB): one counter
// saturates at ~90 appends/us (MI355X_MICROARCH.md, row "dequeue"), which 15k waves would all hit.
constexpr int NSUB = 256;
constexpr int CNT_STRIDE = 16;          // one 128-byte line per counter
constexpr int CNT_TICKET = 1;           // word 1 of a counter's line: the next chunk of that sub-queue the narrowphase hands out

// Items are routed by the kind class of their pair (vp_cls: box-box, box-cylinder, cylinder-cylinder, the rest):
// class c owns cls_groups[c] of the NSUB sub-queues (in proportion to its pairs), a block appends to the (block % groups)-th.  The chunks k_narrow takes are then kind-homogeneous -- one core layout, one
// support routine per side -- which is worth 10 % of its time; one atomicAdd per class present, issued together by lanes 0-3.
NBK_DEV void flush_items(const DevModel& m, unsigned* lds_queue, int qn, int64_t base_cfg, unsigned long long* q_count,
                         unsigned long long* q_items, unsigned long long cap_sub, int lane, unsigned char* ovf) {
    __syncthreads();
    for (int i0 = 0; i0 < qn; i0 += WAVE) {
        const int i = i0 + lane;
        const bool has = i < qn;
        const unsigned it = has ? lds_queue[i] : 0u;
        const int cls = has ? m.vp_cls[it >> 6] : 0;
        unsigned long long bc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bc[c] = __builtin_amdgcn_ballot_w64(has && cls == c);
        const unsigned long long mine_cnt = lane == 0 ? bc[0] : (lane == 1 ? bc[1] : (lane == 2 ? bc[2] : bc[3]));
        unsigned long long off = 0;
        if (lane < 4 && mine_cnt != 0ull)
            off = atomicAdd(q_count + (size_t)(m.cls_base[lane] + (int)(blockIdx.x % (unsigned)m.cls_groups[lane])) * CNT_STRIDE,
                            (unsigned long long)__builtin_popcountll(mine_cnt));
        const unsigned olo = (unsigned)__builtin_amdgcn_ds_bpermute(cls * 4, (int)(unsigned)off);
        const unsigned ohi = (unsigned)__builtin_amdgcn_ds_bpermute(cls * 4, (int)(unsigned)(off >> 32));
        if (has) {
            const unsigned long long mb_ = cls == 0 ? bc[0] : (cls == 1 ? bc[1] : (cls == 2 ? bc[2] : bc[3]));
            const unsigned long long slot = (((unsigned long long)ohi << 32) | olo) +
                                            __builtin_amdgcn_mbcnt_hi((unsigned)(mb_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb_, 0u));
            const unsigned sub = (unsigned)(m.cls_base[cls] + (int)(blockIdx.x % (unsigned)m.cls_groups[cls]));
            const unsigned long long b = (unsigned long long)(base_cfg + (it & 63u));
            if (slot < cap_sub) q_items[(unsigned long long)sub * cap_sub + slot] = (b << 20) | (unsigned long long)(it >> 6);
            else if (ovf != nullptr) ovf[blockIdx.x] = 1;          // the sub-queue is full: this block is re-decided without a queue
        }
    }
    __syncthreads();
}

// the NSUB queue counters are cleared by a kernel of our own: a hipMemsetAsync node did not reliably clear them
// when the call was replayed from a captured hipGraph (ROCm 7.2), a plain kernel node does
__global__ void k_zero_counters(unsigned long long* __restrict__ q_count, unsigned long long* __restrict__ flags, int flag_words) {
    q_count[threadIdx.x * CNT_STRIDE] = 0ull;
    q_count[threadIdx.x * CNT_STRIDE + CNT_TICKET] = 0ull;
    for (int i = threadIdx.x; i < flag_words; i += NSUB) flags[i] = 0ull;
}

// Per-call tables of the float32 broadphase, built once per launch sequence by one workgroup (instead of by every wave
// into LDS) and read by k_broad_f32 through scalar loads: everything in them is uniform over the launch.
//   rkey[16*16] float  radius sum (rounded up) of robot-robot slot (a,b), a < b; < 0 = not a pair
//   rp  [16*16] int    sorted pair index of the slot
//   wkey[W*16], wtc[W*16] float, wp[W*16] int   the same for (world w, robot a); planes: key = t rounded up
//   rho [16]    float  bounding radius of robot shape a, rounded up
//   rcert[16*16], wcert[W*16] float  (thr + inA + inB - f_e2max)^2: centres closer than that => the balls inscribed in the
//               two shapes overlap => the pair collides, whatever the narrowphase would say in more digits; planes: thr + inA
//   wcin [W*16] float  world boxes: (thr + inA + mBox - f_e2max)^2 against the centre's squared distance to the (core) box
//   rptri[128]  int    sorted pair index of robot-robot slot (a,b) at its triangular index b(b-1)/2 + a
//   wlist[W], n_reach  int   the world shapes some robot shape can reach at this threshold, ascending; the world loop visits only those
//   rneg[16*16], rnd[16*16] float  robot-robot slots of the fast stage: e = |ca - cb|^2 + rneg (an fma chain that starts from rneg =
//               -(rs + f_e2max)^2, the candidate threshold with the STATIC slack folded in, squared and rounded up) is negative for a
//               candidate, f = e + rnd (rnd = that threshold minus the certification threshold) for a certified hit; +1 / +3e38 = never
//   wbx[W][8][12] float  world BOXES, fast stage, per pair of robot shapes (2i, 2i + 1): (wkey2 | cull2 | cin | kin | g) x 2 -- the
//               sphere threshold, ((tc+ + rhoA) + f_e2max)^2 for the centre's squared distance to the core box (farther: free),
//               the certification thresholds outside / inside, the depth the centre must exceed inside; see the kernel
//   wkey2[W*16] float  the candidate thresholds of the world slots with the static slack folded in and squared ((rs + f_e2max)^2; planes: the
//               height rs + rhoA + f_e2max): scalars for the waves whose lanes all stay within the static slack
NBK_DEV size_t ftab_entries(int W) { return 5 * 256 + 128 + 6 * (size_t)W * 16 + 32 + (size_t)W + 16 + 96 * (size_t)W + 16; }
struct FTab {
    const float *rkey, *rcert, *wkey, *wtc, *wcert, *wcin, *rho, *rneg, *rnd, *wkey2, *wbx;
    const int *rp, *rptri, *wp, *wlist, *n_reach;
};
NBK_DEV FTab ftab_view(const float* tab, int W) {
    FTab t;
    const size_t w16 = (size_t)W * 16;
    // every [.][16] table starts on a multiple of 16 floats from `tab` (itself 64-byte aligned): a row is one s_load_dwordx16
    t.rkey = tab; t.rp = reinterpret_cast<const int*>(tab + 256); t.rcert = tab + 512;
    t.rptri = reinterpret_cast<const int*>(tab + 768);
    t.rneg = tab + 896; t.rnd = tab + 1152;
    t.wkey = tab + 1408; t.wtc = t.wkey + w16; t.wp = reinterpret_cast<const int*>(t.wtc + w16);
    t.wcert = t.wtc + 2 * w16; t.wcin = t.wcert + w16; t.wkey2 = t.wcin + w16; t.rho = t.wkey2 + w16;
    t.n_reach = reinterpret_cast<const int*>(t.rho + 32); t.wlist = t.n_reach + 16;
    t.wbx = reinterpret_cast<const float*>(t.wlist + ((W + 15) & ~15));        // (16-float aligned)
    return t;
}

__global__ __launch_bounds__(256) void k_prepare_f32(DevModel m, double thr, unsigned long long* __restrict__ q_count, int n_sets, float* __restrict__ tab,
                                                     unsigned long long* __restrict__ flags, int flag_words) {
    const int t = threadIdx.x;
    const int W = m.n_wshapes;
    for (int s = 0; s < n_sets; ++s) { q_count[((size_t)s * NSUB + t) * CNT_STRIDE] = 0ull; q_count[((size_t)s * NSUB + t) * CNT_STRIDE + CNT_TICKET] = 0ull; }
    for (int i = t; i < flag_words; i += 256) flags[i] = 0ull;           // overflow marks of the tile's blocks
    const FTab v = ftab_view(tab, W);
    float* rkey = const_cast<float*>(v.rkey); float* rcert = const_cast<float*>(v.rcert);
    int* rp = const_cast<int*>(v.rp); int* rptri = const_cast<int*>(v.rptri);
    float* wkey = const_cast<float*>(v.wkey); float* wtc = const_cast<float*>(v.wtc); int* wp = const_cast<int*>(v.wp);
    float* wcert = const_cast<float*>(v.wcert); float* wcin = const_cast<float*>(v.wcin); float* rho = const_cast<float*>(v.rho);
    const float up = 1.0f + 2.4e-7f, dn = 1.0f - 2.4e-7f;
    // squared certification threshold: (c - static slack)^2 rounded down, -1 (never) when that is not positive
    auto cert2 = [&](double c) {
        const float v = (float)c * dn - m.f_e2max * up;
        return v > 0.0f ? (v * v) * (dn * dn) : -1.0f;
    };
    float* rneg = const_cast<float*>(v.rneg); float* rnd = const_cast<float*>(v.rnd); float* wkey2 = const_cast<float*>(v.wkey2);
    // squared candidate threshold with the static slack: ((rs rounded up) + f_e2max)^2 rounded up; -1 = not a pair
    auto cand2 = [&](float rs_up) { const float r = (rs_up + m.f_e2max) * up; return (r * r) * (up * up); };
    rkey[t] = -1.0f; rp[t] = -1; rcert[t] = -1.0f; rneg[t] = 1.0f; rnd[t] = 3.0e38f;
    if (t < 128) rptri[t] = -1;
    for (int i = t; i < W * 16; i += 256) { wkey[i] = -1.0f; wtc[i] = 0.0f; wp[i] = -1; wcert[i] = -3.0e38f; wcin[i] = -1.0f; wkey2[i] = -3.0e38f; }
    float* wbx = const_cast<float*>(v.wbx);
    for (int i = t; i < W * 96; i += 256) { const int f = (i % 12) / 2; wbx[i] = f == 0 ? -3.0e38f : (f == 1 ? 0.0f : (f == 4 ? 3.0e38f : -1.0f)); }
    if (t < 16) rho[t] = t < m.n_rshapes ? (float)m.rs_core[6 * t + 5] * up : 0.0f;
    __syncthreads();
    const int P = (NBK_DBG(m) & 2) ? 0 : m.n_pairs;
    for (int j = t; j < P; j += 256) {
        const int* bt = m.bq_tab + 4 * j;
        const int a = bt[0] / 3, p = bt[2], cat = bt[3];
        const double* cst = m.vp_cst + 4 * p;
        if (cat == 1) {
            const int b = bt[1] / 3;
            const double tc = (thr + cst[0]) + cst[1];
            const double rs = (tc + cst[2]) + cst[3];
            const int lo = a < b ? a : b, hi = a < b ? b : a;
            rkey[lo * 16 + hi] = rs > 0.0 ? (float)rs * up : -1.0f;
            rp[lo * 16 + hi] = p;
            rptri[hi * (hi - 1) / 2 + lo] = p;
            const float ce = cert2((thr + m.rs_in[a]) + m.rs_in[b]);
            rcert[lo * 16 + hi] = ce;
            // the fast stage evaluates e = fma(dz, dz, fma(dy, dy, fma(dx, dx, -k2))) and f = e + (k2 - ce'): four roundings of at
            // most 2^-24 max(k2, |d|^2) each.  k2 carries 1e-6 of itself on top of the candidate threshold (a cull e >= 0 still
            // implies |d|^2 >= that threshold), ce' gives up 1e-6 k2 (f < 0 still implies |d|^2 < ce); k2 - ce' is rounded up
            if (rs > 0.0) {
                const float k2 = cand2((float)rs * up) * (1.0f + 1.0e-6f);
                rneg[lo * 16 + hi] = -k2;
                const float cea = ce > 0.0f ? ce * dn - 1.0e-6f * k2 : -1.0f;
                rnd[lo * 16 + hi] = cea > 0.0f ? (k2 - cea) * up : 3.0e38f;
            }
        } else {
            const int w = bt[1];
            // static reach culling: this world shape is out of the shape's reach for this threshold, whatever q is
            const double sl = m.bq_static[j];
            const double tcut = cat == 0 ? thr + cst[0] : (thr + cst[0]) + cst[1];
            if (sl - 1e-9 * (1.0 + __builtin_fabs(sl)) >= tcut) continue;
            float key;
            if (cat == 0) {
                const double tt = thr + cst[0];
                key = (float)tt + __builtin_fabsf((float)tt) * 2.4e-7f;
                // planes: a height, not a squared distance: hc < (thr + inA) - slack
                const float ck = (float)(thr + m.rs_in[a]);
                wcert[w * 16 + a] = (ck - __builtin_fabsf(ck) * 2.4e-7f) - m.f_e2max * up;
            } else {
                const double tc = (thr + cst[0]) + cst[1];
                const double rs = (tc + cst[2]) + cst[3];
                key = rs > 0.0 ? (float)rs * up : -1.0f;
                wtc[w * 16 + a] = (float)tc;
                wcert[w * 16 + a] = cert2((thr + m.rs_in[a]) + m.ws_in[w]);
                // world box: the subject's inscribed ball against the core box; which of cst[0] / cst[1] is the box's margin
                // depends on the user order of the pair -- the box is always the world shape = the second of the pair
                wcin[w * 16 + a] = cert2((thr + m.rs_in[a]) + cst[1]);
            }
            wkey[w * 16 + a] = key;
            // planes: candidate iff hc - rhoA < key + slack  <=>  hc < key + rhoA + slack
            wkey2[w * 16 + a] = cat == 0 ? ((key + rho[a]) + m.f_e2max * up) + __builtin_fabsf(key + rho[a]) * 2.4e-7f
                                         : (key >= 0.0f ? cand2(key) : -1.0f);
            wp[w * 16 + a] = p;
            if (cat != 0 && m.ws_kind[w] == K_HULL) {
                // the fast stage's hull slot: bounding spheres, the cull against the hull's local box, the inscribed balls
                float* wb = wbx + w * 96 + (a / 2) * 12 + (a % 2);
                const double tc = (thr + cst[0]) + cst[1];
                const float tcp = tc > 0.0 ? (float)tc * up : 0.0f;
                const float rr = ((tcp + rho[a]) + m.f_e2max * up) * up;
                wb[0] = wkey2[w * 16 + a];
                wb[2] = (rr * rr) * (up * up);
                wb[4] = wcert[w * 16 + a];
            }
            if (cat != 0 && m.ws_kind[w] == K_BOX) {
                // the fast stage's box slot (every threshold with the STATIC slack bound, so that it is a scalar): see k_broad_f32
                float* wb = wbx + w * 96 + (a / 2) * 12 + (a % 2);
                const double tc = (thr + cst[0]) + cst[1];
                const float tcp = tc > 0.0 ? (float)tc * up : 0.0f;
                const float rr = ((tcp + rho[a]) + m.f_e2max * up) * up;
                wb[0] = wkey2[w * 16 + a];
                wb[2] = (rr * rr) * (up * up);                                   // centre farther from the core box than this: free
                wb[4] = wcin[w * 16 + a];                                        // closer than this (and outside): certain hit
                wb[6] = key > m.f_e2max * up ? ((key - m.f_e2max * up) * (key - m.f_e2max * up)) * (dn * dn * dn) : -1.0f;   // inside + within this sphere ...
                wb[8] = ((float)(-tc) + __builtin_fabsf((float)tc) * 2.4e-7f) + m.f_e2max * up;                               // ... + deeper than this: certain hit
            }
        }
    }
    // ---- the world shapes that are still somebody's pair, in ascending order ---------------------------------------------
    __shared__ int s_cnt[4];
    __shared__ int s_run;
    if (t == 0) s_run = 0;
    __syncthreads();                       // (also orders the wp[] writes above before the reads below)
    int* wlist = const_cast<int*>(v.wlist);
    for (int w0 = 0; w0 < W; w0 += 256) {
        const int w = w0 + t;
        bool reach = false;
        if (w < W) for (int a = 0; a < 16; ++a) reach = reach || (wp[w * 16 + a] >= 0);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(reach);
        const int wave = t >> 6;
        if ((t & 63) == 0) s_cnt[wave] = __builtin_popcountll(bal);
        __syncthreads();
        int off = s_run;
        for (int i = 0; i < wave; ++i) off += s_cnt[i];
        if (reach) wlist[off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = w;
        __syncthreads();
        if (t == 0) s_run += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (t == 0) *const_cast<int*>(v.n_reach) = s_run;
}

// LDS: raw q slab [64*n_q] | saved frames [12*slots][64] | centres [3*S][64] | pair constants [P][4] |
//      world cores [W][18] | queue [BQ_CAP] u32.
// Lane = configuration throughout.  Everything a pair needs that does not depend on the configuration
// (rows, squared bounding radii for THIS threshold, box constants) is put into LDS once per wave and read
// back with wave-uniform addresses (broadcast reads), so the pair loop touches no scalar or vector memory
// and is a straight line per category.  Survivors are collected as one bit per (lane, pair) and turned into
// queue items 64 pairs at a time.
NBK_DEV void enqueue_bits(const DevModel& m, unsigned long long bits, int jbase, const double* lds_pc, unsigned* lds_queue, int& qn, int lane,
                          int64_t base, unsigned long long* q_count, unsigned long long* q_items, unsigned long long cap, const EdgeSrc& es) {
    while (true) {
        const bool has = bits != 0ull;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
        if (bal == 0ull) break;
        if (has) {
            const int bit = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            const unsigned long long e0 = __builtin_bit_cast(unsigned long long, lds_pc[4 * (jbase + bit)]);
            const unsigned p = (unsigned)(e0 >> 32) & 0xFFFFFu;
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            lds_queue[pos] = (p << 6) | (unsigned)lane;
        }
        qn += __builtin_popcountll(bal);
        if (qn > BQ_CAP - WAVE) { flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf); qn = 0; }
    }
}


__global__ __launch_bounds__(64) void k_broad(DevModel m, EdgeSrc es, const double* __restrict__ q, int64_t B, double thr,
                                               uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
                                               unsigned long long* __restrict__ q_count, unsigned long long* __restrict__ q_items,
                                               unsigned long long cap) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nq = m.n_q;
    const int P = (NBK_DBG(m) & 2) ? 0 : m.n_pairs;
    double* lds_raw = lds;
    const int qrows = (WAVE * nq * 8 >= BQ_CAP * 4) ? nq : (BQ_CAP * 4 + WAVE * 8 - 1) / (WAVE * 8);
    double* lds_fr = lds_raw + WAVE * qrows;
    double* lds_c = lds_fr + WAVE * 12 * m.frame_slots;
    double* lds_pc = lds_c + WAVE * 3 * m.n_rshapes;
    double* lds_w = lds_pc + 4 * m.n_pairs;
    // the queue reuses the q slab, which is dead once the sweep is over (host sizes the region for both)
    unsigned* lds_queue = reinterpret_cast<unsigned*>(lds_raw);
    const int64_t Beff = effective_batch(es, B);
    if (base >= Beff) return;               // edge mode: the launch covers the scratch's capacity, this block lies beyond the samples
    const int rows_i = (int)((Beff - base) < WAVE ? (Beff - base) : WAVE);
    // ---- stage q (coalesced), no transposed copy: lane reads lds_raw[lane*nq + j] -------------------------
    if (es.map != nullptr) {
        // edge mode: this lane's configuration is an interpolation sample
        if (lane < rows_i) {
            unsigned e;
            const double t = edge_t(es, es.map[base + lane], e);
            const double omt = 1.0 - t;
            const double* sp = es.starts + (size_t)e * nq;
            const double* gp = es.goals + (size_t)e * nq;
            for (int j = 0; j < nq; ++j) { const double a = omt * sp[j]; const double bb = t * gp[j]; lds_raw[lane * nq + j] = a + bb; }
        } else {
            for (int j = 0; j < nq; ++j) lds_raw[lane * nq + j] = 0.0;
        }
    } else {
        const int total = rows_i * nq;
        const double* src = q + base * nq;
#if defined(NBK_BF32_ABL) && NBK_BF32_ABL == 3      // timing experiment: no global read of q
        if (true) { for (int i = lane; i < WAVE * nq; i += WAVE) lds_raw[i] = 0.001 * (double)(i + (int)blockIdx.x); } else
#endif
        if (rows_i == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(lds_raw);
            for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += WAVE) lds_raw[i] = src[i];
            for (int i = total + lane; i < WAVE * nq; i += WAVE) lds_raw[i] = 0.0;
        }
    }
    // ---- per-wave constants: lane j prepares pair j ------------------------------------------------------------
    for (int j = lane; j < P; j += WAVE) {
        const int* t = m.bq_tab + 4 * j;
        const int p = t[2], cat = t[3];
        const double* cst = m.vp_cst + 4 * p;
        const unsigned long long e0 = (unsigned long long)(unsigned)(t[0] & 0xFFFF) | ((unsigned long long)(unsigned)(t[1] & 0xFFFF) << 16) |
                                      ((unsigned long long)(unsigned)p << 32) | ((unsigned long long)(unsigned)cat << 60);
        double key, tc = 0.0;
        if (cat == 0) key = thr + cst[0];
        else {
            tc = (thr + cst[0]) + cst[1];
            const double rs = (tc + cst[2]) + cst[3];
            key = rs > 0.0 ? rs * rs : -1.0;
        }
        lds_pc[4 * j] = __builtin_bit_cast(double, e0);
        lds_pc[4 * j + 1] = key;
        lds_pc[4 * j + 2] = tc;
        lds_pc[4 * j + 3] = cst[2];
    }
    for (int i = lane; i < 18 * m.n_wshapes; i += WAVE) lds_w[i] = m.ws_core[i];
    __syncthreads();
    const bool active = lane < rows_i;
    bool hit = row_nonfinite(lds_raw + lane * nq, nq, 1);        // non-finite joint values: colliding, nothing queued
    // ---- sweep: centres only ----------------------------------------------------------------------------------
    {
        Xf bpose;
        xf_from12(m.base_pose, bpose);
        Xf T = bpose;
        for (int k = -1; k < m.n_joints; ++k) {
            if (k >= 0) {
                const int ld = m.joint_load[k];
                Xf Pf;
                if (ld == -2) Pf = T;
                else if (ld == -1) Pf = bpose;
                else {
#pragma unroll
                    for (int e = 0; e < 9; ++e) Pf.R[e] = lds_fr[(ld * 12 + e) * WAVE + lane];
#pragma unroll
                    for (int e = 0; e < 3; ++e) Pf.t[e] = lds_fr[(ld * 12 + 9 + e) * WAVE + lane];
                }
                const double qk = lds_raw[lane * nq + m.joint_qidx[k]];
                joint_apply(m, k, Pf, qk, T);
                const int sv = m.joint_save[k];
                if (sv >= 0) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) lds_fr[(sv * 12 + e) * WAVE + lane] = T.R[e];
#pragma unroll
                    for (int e = 0; e < 3; ++e) lds_fr[(sv * 12 + 9 + e) * WAVE + lane] = T.t[e];
                }
            }
            const int s0 = m.joint_shape_begin[k + 1], s1 = m.joint_shape_begin[k + 2];
            for (int s = s0; s < s1; ++s) {
                const double* loc = m.rs_local + 12 * s;
                const double tl[3] = {loc[3], loc[7], loc[11]};
                double c[3];
                xf_mul_pos(T, tl, c);
                double* rows = lds_c + (3 * s) * WAVE + lane;
                rows[0] = c[0]; rows[WAVE] = c[1]; rows[2 * WAVE] = c[2];
            }
        }
    }
    __syncthreads();            // the q slab is dead from here on: its LDS region becomes the item queue
    // ---- broadphase: one bit per surviving (lane, pair) -------------------------------------------------------------
    // Pairs are category-major; each category runs a branch-free body, unrolled so that the LDS reads of several
    // pairs are in flight before the first use (the loop is latency-, not throughput-bound otherwise).
    int qn = 0;
    int j0 = 0;
    for (int cat = 0; cat < 4; ++cat) {
        const int ncat = (NBK_DBG(m) & 2) ? 0 : m.bq_count[cat];
        for (int c0 = 0; c0 < ncat; c0 += WAVE) {
            const int nn = (ncat - c0) < WAVE ? (ncat - c0) : WAVE;
            const double* pc0 = lds_pc + 4 * (j0 + c0);
            unsigned long long bits = 0ull;
            if (cat == 1) {
#pragma unroll 4
                for (int u = 0; u < nn; ++u) {
                    const unsigned long long e0 = __builtin_bit_cast(unsigned long long, pc0[4 * u]);
                    const double key = pc0[4 * u + 1];
                    const double* ra = lds_c + (int)(e0 & 0xFFFFull) * WAVE + lane;
                    const double* rb = lds_c + (int)((e0 >> 16) & 0xFFFFull) * WAVE + lane;
                    const double d[3] = {ra[0] - rb[0], ra[WAVE] - rb[WAVE], ra[2 * WAVE] - rb[2 * WAVE]};
                    bits |= (dot3(d, d) < key) ? (1ull << u) : 0ull;
                }
            } else if (cat == 2) {
#pragma unroll 4
                for (int u = 0; u < nn; ++u) {
                    const unsigned long long e0 = __builtin_bit_cast(unsigned long long, pc0[4 * u]);
                    const double key = pc0[4 * u + 1];
                    const double* ra = lds_c + (int)(e0 & 0xFFFFull) * WAVE + lane;
                    const double* wc = lds_w + 18 * (int)((e0 >> 16) & 0xFFFFull);
                    const double d[3] = {ra[0] - wc[0], ra[WAVE] - wc[1], ra[2 * WAVE] - wc[2]};
                    bits |= (dot3(d, d) < key) ? (1ull << u) : 0ull;
                }
            } else if (cat == 3) {
#pragma unroll 2
                for (int u = 0; u < nn; ++u) {
                    const unsigned long long e0 = __builtin_bit_cast(unsigned long long, pc0[4 * u]);
                    const double key = pc0[4 * u + 1], tcb = pc0[4 * u + 2], rhoA = pc0[4 * u + 3];
                    const double* ra = lds_c + (int)(e0 & 0xFFFFull) * WAVE + lane;
                    const double* wc = lds_w + 18 * (int)((e0 >> 16) & 0xFFFFull);
                    const double ca[3] = {ra[0], ra[WAVE], ra[2 * WAVE]};
                    const double d[3] = {ca[0] - wc[0], ca[1] - wc[1], ca[2] - wc[2]};
                    Core bx;
                    bx.kind = K_BOX;
#pragma unroll
                    for (int e = 0; e < 3; ++e) { bx.c[e] = wc[e]; bx.ax[0][e] = wc[3 + e]; bx.ax[1][e] = wc[6 + e]; bx.ax[2][e] = wc[9 + e]; bx.h[e] = wc[12 + e]; }
                    bx.rad = 0.0; bx.margin = 0.0; bx.rho = 0.0;
                    const int v = box_midphase(ca, rhoA, bx, tcb);
                    const bool cand = (dot3(d, d) < key) && (v != 0);
                    hit = hit || (cand && v == 1);
                    bits |= cand ? (1ull << u) : 0ull;
                }
            } else {
                for (int u = 0; u < nn; ++u) {
                    const unsigned long long e0 = __builtin_bit_cast(unsigned long long, pc0[4 * u]);
                    const double key = pc0[4 * u + 1], rhoA = pc0[4 * u + 3];
                    const double* ra = lds_c + (int)(e0 & 0xFFFFull) * WAVE + lane;
                    const double* wc = lds_w + 18 * (int)((e0 >> 16) & 0xFFFFull);
                    const double d[3] = {ra[0] - wc[0], ra[WAVE] - wc[1], ra[2 * WAVE] - wc[2]};
                    const double n[3] = {wc[9], wc[10], wc[11]};
                    bits |= !((dot3(d, n) - rhoA) >= key) ? (1ull << u) : 0ull;
                }
            }
            if (!active || hit || (NBK_DBG(m) & 4)) bits = 0ull;
            enqueue_bits(m, bits, j0 + c0, lds_pc, lds_queue, qn, lane, base, q_count, q_items, cap, es);
        }
        j0 += ncat;
    }
    if (qn > 0) flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf);
    // the mask starts from the hits certified here; k_narrow ORs the rest in
    const unsigned long long word = __builtin_amdgcn_ballot_w64(hit && active);
    if (mask_bits != nullptr && lane == 0) mask_bits[blockIdx.x] = word;
    if (mask_bytes != nullptr && active) mask_bytes[base + lane] = hit ? 1 : 0;
}

// ---- k_broad_reg<S>: the same broadphase with the centres of up to S robot shapes in REGISTERS ---------------
// The shape and pair loops are unrolled at compile time (S is a template parameter; which (a, b) slots are real
// pairs is run-time data), so cx[a] / cx[b] are plain registers: no LDS traffic in the pair tests and no LDS
// budget for centres -- occupancy is set by VGPRs (~3 waves/SIMD) instead of by 17 KB of LDS per wave.
// LDS: raw q slab (later the item queue) | saved frames | key / pair-index tables [S*S] and [W*S].
template <int S>
__global__ __launch_bounds__(64, 3) void k_broad_reg(DevModel m, EdgeSrc es, const double* __restrict__ q, int64_t B, double thr,
                                                   uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
                                                   unsigned long long* __restrict__ q_count, unsigned long long* __restrict__ q_items,
                                                   unsigned long long cap) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nq = m.n_q;
    const int W = m.n_wshapes;
    double* lds_raw = lds;
    const int qrows = (WAVE * nq * 8 >= BQ_CAP * 4) ? nq : (BQ_CAP * 4 + WAVE * 8 - 1) / (WAVE * 8);
    double* lds_fr = lds_raw + WAVE * qrows;
    double* lds_rkey = lds_fr + WAVE * 12 * m.frame_slots;      // [S*S] key of robot-robot slot (a,b), a < b; < 0 = no pair
    double* lds_wkey = lds_rkey + S * S;                         // [W*S] key of (world w, robot a)
    double* lds_wtc = lds_wkey + W * S;                          // [W*S] tc of (world box w, robot a)
    int* lds_rp = reinterpret_cast<int*>(lds_wtc + W * S);       // [S*S] sorted pair index of the slot
    int* lds_wp = lds_rp + S * S;                                // [W*S]
    unsigned* lds_queue = reinterpret_cast<unsigned*>(lds_raw);
    const int64_t Beff = effective_batch(es, B);
    if (base >= Beff) return;               // edge mode: the launch covers the scratch's capacity, this block lies beyond the samples
    const int rows_i = (int)((Beff - base) < WAVE ? (Beff - base) : WAVE);
    // ---- stage q ------------------------------------------------------------------------------------------------------
    if (es.map != nullptr) {
        if (lane < rows_i) {
            unsigned e;
            const double t = edge_t(es, es.map[base + lane], e);
            const double omt = 1.0 - t;
            const double* sp = es.starts + (size_t)e * nq;
            const double* gp = es.goals + (size_t)e * nq;
            for (int j = 0; j < nq; ++j) { const double a = omt * sp[j]; const double bb = t * gp[j]; lds_raw[lane * nq + j] = a + bb; }
        } else {
            for (int j = 0; j < nq; ++j) lds_raw[lane * nq + j] = 0.0;
        }
    } else {
        const int total = rows_i * nq;
        const double* src = q + base * nq;
#if defined(NBK_BF32_ABL) && NBK_BF32_ABL == 3      // timing experiment: no global read of q
        if (true) { for (int i = lane; i < WAVE * nq; i += WAVE) lds_raw[i] = 0.001 * (double)(i + (int)blockIdx.x); } else
#endif
        if (rows_i == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(lds_raw);
            for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += WAVE) lds_raw[i] = src[i];
            for (int i = total + lane; i < WAVE * nq; i += WAVE) lds_raw[i] = 0.0;
        }
    }
    // ---- per-wave tables: slot (a,b) -> key for THIS threshold and pair index ------------------------------------------------
    for (int i = lane; i < S * S; i += WAVE) { lds_rkey[i] = -1.0; lds_rp[i] = -1; }
    for (int i = lane; i < W * S; i += WAVE) { lds_wkey[i] = -1.0; lds_wtc[i] = 0.0; lds_wp[i] = -1; }
    __syncthreads();
    const int P = (NBK_DBG(m) & 2) ? 0 : m.n_pairs;
    for (int j = lane; j < P; j += WAVE) {
        const int* t = m.bq_tab + 4 * j;
        const int a = t[0] / 3, p = t[2], cat = t[3];
        const double* cst = m.vp_cst + 4 * p;
        if (cat == 1) {
            const int b = t[1] / 3;
            const double tc = (thr + cst[0]) + cst[1];
            const double rs = (tc + cst[2]) + cst[3];
            const int lo = a < b ? a : b, hi = a < b ? b : a;
            lds_rkey[lo * S + hi] = rs > 0.0 ? rs * rs : -1.0;
            lds_rp[lo * S + hi] = p;
        } else {
            const int w = t[1];
            double key;
            if (cat == 0) key = thr + cst[0];
            else {
                const double tc = (thr + cst[0]) + cst[1];
                const double rs = (tc + cst[2]) + cst[3];
                key = rs > 0.0 ? rs * rs : -1.0;
                lds_wtc[w * S + a] = tc;
            }
            lds_wkey[w * S + a] = key;
            lds_wp[w * S + a] = p;
        }
    }
    __syncthreads();
    const bool active = lane < rows_i;
    bool hit = row_nonfinite(lds_raw + lane * nq, nq, 1);        // non-finite joint values: colliding, nothing queued
    // ---- sweep: centres into registers ---------------------------------------------------------------------------------------
    double cx[S], cy[S], cz[S];
    {
        Xf bpose;
        xf_from12(m.base_pose, bpose);
        Xf T = bpose;
        int kcur = -1;
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) {
            cx[sidx] = 0.0; cy[sidx] = 0.0; cz[sidx] = 0.0;
            if (sidx < m.n_rshapes) {
                const int f = m.rs_frame[sidx];
                while (kcur < f) {
                    ++kcur;
                    const int k = kcur;
                    const int ld = m.joint_load[k];
                    Xf Pf;
                    if (ld == -2) Pf = T;
                    else if (ld == -1) Pf = bpose;
                    else {
#pragma unroll
                        for (int e = 0; e < 9; ++e) Pf.R[e] = lds_fr[(ld * 12 + e) * WAVE + lane];
#pragma unroll
                        for (int e = 0; e < 3; ++e) Pf.t[e] = lds_fr[(ld * 12 + 9 + e) * WAVE + lane];
                    }
                    const double qk = lds_raw[lane * nq + m.joint_qidx[k]];
                    joint_apply(m, k, Pf, qk, T);
                    const int sv = m.joint_save[k];
                    if (sv >= 0) {
#pragma unroll
                        for (int e = 0; e < 9; ++e) lds_fr[(sv * 12 + e) * WAVE + lane] = T.R[e];
#pragma unroll
                        for (int e = 0; e < 3; ++e) lds_fr[(sv * 12 + 9 + e) * WAVE + lane] = T.t[e];
                    }
                }
                const double* loc = m.rs_local + 12 * sidx;
                const double tl[3] = {loc[3], loc[7], loc[11]};
                double c[3];
                if (f < 0) xf_mul_pos(bpose, tl, c); else xf_mul_pos(T, tl, c);
                cx[sidx] = c[0]; cy[sidx] = c[1]; cz[sidx] = c[2];
            }
        }
    }
    __syncthreads();            // the q slab is dead from here on: its LDS region becomes the item queue
    int qn = 0;
    // ---- robot-robot slots (static a < b) ----------------------------------------------------------------------------------------
    if (m.bq_count[1] > 0) {
#pragma unroll
        for (int a = 0; a < S - 1; ++a) {
            unsigned long long bits = 0ull;
#pragma unroll
            for (int b = a + 1; b < S; ++b) {
                const double key = lds_rkey[a * S + b];
                const double d[3] = {cx[a] - cx[b], cy[a] - cy[b], cz[a] - cz[b]};
                bits |= (dot3(d, d) < key) ? (1ull << b) : 0ull;
            }
            if (!active || hit) bits = 0ull;
            // enqueue this row's survivors
            while (true) {
                const bool has = bits != 0ull;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
                if (bal == 0ull) break;
                if (has) {
                    const int bit = __builtin_ctzll(bits);
                    bits &= bits - 1ull;
                    const unsigned p = (unsigned)lds_rp[a * S + bit];
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    lds_queue[pos] = (p << 6) | (unsigned)lane;
                }
                qn += __builtin_popcountll(bal);
                if (qn > BQ_CAP - WAVE) { flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf); qn = 0; }
            }
        }
    }
    // ---- world shapes (run-time loop) against the static robot shapes -------------------------------------------------------------
    for (int w = 0; w < W; ++w) {
        const double* wc = m.ws_core + 18 * w;
        const int wk = m.ws_kind[w];
        unsigned long long bits = 0ull;
        if (wk == K_PLANE) {
            const double n[3] = {wc[9], wc[10], wc[11]};
#pragma unroll
            for (int a = 0; a < S; ++a) {
                if (a < m.n_rshapes) {
                    const double key = lds_wkey[w * S + a];
                    const double rhoA = m.rs_core[6 * a + 5];
                    const double d[3] = {cx[a] - wc[0], cy[a] - wc[1], cz[a] - wc[2]};
                    const bool cand = (lds_wp[w * S + a] >= 0) && !((dot3(d, n) - rhoA) >= key);
                    bits |= cand ? (1ull << a) : 0ull;
                }
            }
        } else if (wk == K_BOX) {
            Core bx;
            bx.kind = K_BOX;
#pragma unroll
            for (int e = 0; e < 3; ++e) { bx.c[e] = wc[e]; bx.ax[0][e] = wc[3 + e]; bx.ax[1][e] = wc[6 + e]; bx.ax[2][e] = wc[9 + e]; bx.h[e] = wc[12 + e]; }
            bx.rad = 0.0; bx.margin = 0.0; bx.rho = 0.0;
#pragma unroll
            for (int a = 0; a < S; ++a) {
                if (a < m.n_rshapes) {
                    const double key = lds_wkey[w * S + a];
                    const double ca[3] = {cx[a], cy[a], cz[a]};
                    const double d[3] = {ca[0] - wc[0], ca[1] - wc[1], ca[2] - wc[2]};
                    const int v = box_midphase(ca, m.rs_core[6 * a + 5], bx, lds_wtc[w * S + a]);
                    const bool cand = (dot3(d, d) < key) && (v != 0);
                    hit = hit || (cand && v == 1);
                    bits |= cand ? (1ull << a) : 0ull;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < S; ++a) {
                if (a < m.n_rshapes) {
                    const double key = lds_wkey[w * S + a];
                    const double d[3] = {cx[a] - wc[0], cy[a] - wc[1], cz[a] - wc[2]};
                    bits |= (dot3(d, d) < key) ? (1ull << a) : 0ull;
                }
            }
        }
        if (!active || hit || (NBK_DBG(m) & 4)) bits = 0ull;
        while (true) {
            const bool has = bits != 0ull;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
            if (bal == 0ull) break;
            if (has) {
                const int bit = __builtin_ctzll(bits);
                bits &= bits - 1ull;
                const unsigned p = (unsigned)lds_wp[w * S + bit];
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                lds_queue[pos] = (p << 6) | (unsigned)lane;
            }
            qn += __builtin_popcountll(bal);
            if (qn > BQ_CAP - WAVE) { flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf); qn = 0; }
        }
    }
    if (qn > 0) flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf);
    const unsigned long long word = __builtin_amdgcn_ballot_w64(hit && active);
    if (mask_bits != nullptr && lane == 0) mask_bits[blockIdx.x] = word;
    if (mask_bytes != nullptr && active) mask_bytes[base + lane] = hit ? 1 : 0;
}

// ---- k_broad_f32<S>: the register broadphase in float32, conservative -----------------------------------------------
// The broadphase only has to cull pairs that are certainly free; every survivor is decided exactly (float64) by
// k_narrow, which repeats the bounding-sphere test of the predicate in float64 first.  So the sweep, the centres and
// the tests run in float32 (twice the VALU rate, half the registers: 5+ waves per SIMD) with every comparison pushed
// towards "survives" by a slack `e` that covers the float32 error of the sweep 50 times over:
//   spheres  cull  iff |cA - cB|^2_f32 >= ((tc + rhoA + rhoB) + 2e)^2 (rounded up);
//   planes   cull  iff hc_f32 - rhoA >= t + 2e;
//   boxes    cull  iff d^2(c, box)_f32 >= ((tc + rho) + 2e)^2 (tc >= 0);  certified hit iff the centre is inside
//            deeper than -tc by more than 2e (then it is inside in float64 as well).
// Tables in LDS as in k_broad_reg (keys in float32, already including the static part of the slack).
struct XfF { float R[9]; float t[3]; };

// sin / cos for the conservative float32 sweep: the hardware's v_sin_f32 / v_cos_f32 on the fractional part of x / 2 pi (five
// instructions where the Cody-Waite + Taylor form took 22).  Measured on the device over |x| <= 64 (tools: profiles/r03_hw_sincos.log):
// absolute error <= 2.7e-7 for |x| <= 3.2 and <= 2.7e-7 + 4e-8 |x| beyond (the rounding of x / 2 pi) -- inside what the slack
// charges per joint (16 ulp = 9.5e-7 for the sweep, 2.4e-7 |q| for the angle, both times 50)
NBK_DEV void sincos_f(float x, float& s, float& c) {
    const float r = x * 0.15915494309189535f;
    const float f = r - __builtin_rintf(r);
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
}

template <int KZ>
NBK_DEV void joint_apply_axis_f(const float* M, const float* toff, const XfF& P, float s, float c, XfF& o) {
    constexpr int U = (KZ + 1) % 3, V = (KZ + 2) % 3;
    float Lu[3], Lv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Lu[r] = __builtin_fmaf(s, M[18 + 3 * r + U], -c * M[9 + 3 * r + U]);
        Lv[r] = __builtin_fmaf(s, M[18 + 3 * r + V], -c * M[9 + 3 * r + V]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float a0 = P.R[3 * i], a1 = P.R[3 * i + 1], a2 = P.R[3 * i + 2];
        o.R[3 * i + U] = __builtin_fmaf(a2, Lu[2], __builtin_fmaf(a1, Lu[1], a0 * Lu[0]));
        o.R[3 * i + V] = __builtin_fmaf(a2, Lv[2], __builtin_fmaf(a1, Lv[1], a0 * Lv[0]));
        o.R[3 * i + KZ] = __builtin_fmaf(a2, M[6 + KZ], __builtin_fmaf(a1, M[3 + KZ], a0 * M[KZ]));
        o.t[i] = __builtin_fmaf(a2, toff[2], __builtin_fmaf(a1, toff[1], __builtin_fmaf(a0, toff[0], P.t[i])));
    }
}

// the float32 sweep of the conservative broadphase (its error is covered by the slack, so nothing here has to match anything bit for
// bit): same case split as joint_apply
NBK_DEV void joint_apply_f(const DevModel& m, int k, const XfF& P, float qk, XfF& o) {
    const float* M = m.f_tab + 27 * k;
    const float* toff = m.f_tab + m.f_trans + 3 * k;
    const float* sl = m.f_tab + m.f_slide + 3 * k;
    const int kind = m.joint_kind[k];
    float s = 0.0f, c = 0.0f;
    if (kind != JK_PRISMATIC) sincos_f(qk, s, c);
    if (kind == 0) { joint_apply_axis_f<0>(M, toff, P, s, c, o); return; }
    if (kind == 1) { joint_apply_axis_f<1>(M, toff, P, s, c, o); return; }
    if (kind == 2) { joint_apply_axis_f<2>(M, toff, P, s, c, o); return; }
    float L[9], tl[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) L[e] = __builtin_fmaf(s, M[18 + e], __builtin_fmaf(-c, M[9 + e], M[e]));
#pragma unroll
    for (int i = 0; i < 3; ++i) tl[i] = __builtin_fmaf(qk, sl[i], toff[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.R[3 * i + j] = __builtin_fmaf(P.R[3 * i + 2], L[6 + j], __builtin_fmaf(P.R[3 * i + 1], L[3 + j], P.R[3 * i] * L[j]));
        o.t[i] = __builtin_fmaf(P.R[3 * i + 2], tl[2], __builtin_fmaf(P.R[3 * i + 1], tl[1], __builtin_fmaf(P.R[3 * i], tl[0], P.t[i])));
    }
}

struct alignas(64) Row16f { float v[16]; };     // one row of a [.][16] slot table: a single s_load_dwordx16
typedef float V2f __attribute__((ext_vector_type(2)));
typedef int V2i __attribute__((ext_vector_type(2)));
// |d|^2 + nk as one fma chain, for two slots at a time (v_pk_fma_f32) and for one: the same operations in the same order, so the
// one-slot form reproduces the pair form's value bit for bit (the rare enqueue path re-evaluates what the row's sign bits flagged)
NBK_DEV V2f slot_e2(V2f dx, V2f dy, V2f dz, V2f nk) {
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, nk)));
}
NBK_DEV float slot_e1(float dx, float dy, float dz, float nk) { return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, nk))); }

// ---- the packed float32 sweep of k_broad_f32 --------------------------------------------------------------------------------
// A frame as rows 0 and 1 of every column in one register PAIR (Rc[k] = (R[0][k], R[1][k]), tc = (t[0], t[1])) and row 2 apart:
// R x L and R x v then run rows 0 and 1 together on v_pk_fma_f32 (one issue slot for two multiply-adds; plain float32 and packed
// float32 instructions issue at the same rate on this SIMD, profiles/r03_valu_issue_rate.log): 27 instructions per axis-aligned
// joint instead of 48, 6 per shape centre instead of 9.
struct XfP { V2f Rc[3]; float R2[3]; V2f tc; float t2; };
// host-made constants of one joint (f_tab + f_pk + 20 k): for a joint about coordinate axis KZ of its frame, U = KZ + 1, V = KZ + 2
struct alignas(16) JPk { float m2p[6]; float m1p[6]; float mk[3]; float pad0; float toff[3]; float pad1; };
NBK_DEV V2f splat2(float x) { return V2f{x, x}; }
NBK_DEV V2f fma2(V2f a, V2f b, V2f c) { return __builtin_elementwise_fma(a, b, c); }

template <int KZ>
NBK_DEV void joint_apply_axis_p(const JPk& jp, const XfP& P, float s, float c, XfP& o) {
    constexpr int U = (KZ + 1) % 3, V = (KZ + 2) % 3;
    const V2f s2 = splat2(s), c2 = splat2(c);
    V2f Lp[3];                                                     // (L[r][U], L[r][V]) = s M2 - c M1
#pragma unroll
    for (int r = 0; r < 3; ++r) Lp[r] = fma2(s2, V2f{jp.m2p[2 * r], jp.m2p[2 * r + 1]}, -(c2 * V2f{jp.m1p[2 * r], jp.m1p[2 * r + 1]}));
    const V2f cu = fma2(P.Rc[2], splat2(Lp[2].x), fma2(P.Rc[1], splat2(Lp[1].x), P.Rc[0] * splat2(Lp[0].x)));
    const V2f cv = fma2(P.Rc[2], splat2(Lp[2].y), fma2(P.Rc[1], splat2(Lp[1].y), P.Rc[0] * splat2(Lp[0].y)));
    const V2f ck = fma2(P.Rc[2], splat2(jp.mk[2]), fma2(P.Rc[1], splat2(jp.mk[1]), P.Rc[0] * splat2(jp.mk[0])));
    const V2f r2 = fma2(splat2(P.R2[2]), Lp[2], fma2(splat2(P.R2[1]), Lp[1], splat2(P.R2[0]) * Lp[0]));        // row 2, columns U and V
    const float r2k = __builtin_fmaf(P.R2[2], jp.mk[2], __builtin_fmaf(P.R2[1], jp.mk[1], P.R2[0] * jp.mk[0]));
    const V2f tc = fma2(P.Rc[2], splat2(jp.toff[2]), fma2(P.Rc[1], splat2(jp.toff[1]), fma2(P.Rc[0], splat2(jp.toff[0]), P.tc)));
    const float t2 = __builtin_fmaf(P.R2[2], jp.toff[2], __builtin_fmaf(P.R2[1], jp.toff[1], __builtin_fmaf(P.R2[0], jp.toff[0], P.t2)));
    o.Rc[U] = cu; o.Rc[V] = cv; o.Rc[KZ] = ck;
    o.R2[U] = r2.x; o.R2[V] = r2.y; o.R2[KZ] = r2k;
    o.tc = tc; o.t2 = t2;
}

// same case split as joint_apply_f; the error of every form is covered by the slack
NBK_DEV void joint_apply_p(const DevModel& m, int k, int kind, const JPk& jp, const XfP& P, float qk, XfP& o) {
    float s = 0.0f, c = 0.0f;
    if (kind != JK_PRISMATIC) sincos_f(qk, s, c);
    if (kind <= 2) {
        if (kind == 0) joint_apply_axis_p<0>(jp, P, s, c, o);
        else if (kind == 1) joint_apply_axis_p<1>(jp, P, s, c, o);
        else joint_apply_axis_p<2>(jp, P, s, c, o);
        return;
    }
    const float* M = m.f_tab + 27 * k;
    const float* toff = m.f_tab + m.f_trans + 3 * k;
    const float* sl = m.f_tab + m.f_slide + 3 * k;
    float L[9], tl[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) L[e] = __builtin_fmaf(s, M[18 + e], __builtin_fmaf(-c, M[9 + e], M[e]));
#pragma unroll
    for (int i = 0; i < 3; ++i) tl[i] = __builtin_fmaf(qk, sl[i], toff[i]);
    XfP n;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        n.Rc[j] = fma2(P.Rc[2], splat2(L[6 + j]), fma2(P.Rc[1], splat2(L[3 + j]), P.Rc[0] * splat2(L[j])));
        n.R2[j] = __builtin_fmaf(P.R2[2], L[6 + j], __builtin_fmaf(P.R2[1], L[3 + j], P.R2[0] * L[j]));
    }
    n.tc = fma2(P.Rc[2], splat2(tl[2]), fma2(P.Rc[1], splat2(tl[1]), fma2(P.Rc[0], splat2(tl[0]), P.tc)));
    n.t2 = __builtin_fmaf(P.R2[2], tl[2], __builtin_fmaf(P.R2[1], tl[1], __builtin_fmaf(P.R2[0], tl[0], P.t2)));
    o = n;
}

// in-kernel phase timing of k_broad_f32, only in builds made with -DNBK_BF32_STAMP (tools/build_variant.sh, tools/broad_prof.py):
// per wave the cycles between stamps, summed over the launch -- [0] prologue (q slab into LDS), [1] sweep, [2] world shapes,
// [3] robot-robot rows, [4] final flush + mask, [7] waves
#ifdef NBK_BF32_STAMP
__device__ unsigned long long g_broad_prof[16384 * 8];     // one slot per block (mod 16384): plain adds, no contention
#else
__device__ unsigned long long g_broad_prof[8];
#endif
#ifdef NBK_BF32_STAMP
#define NBK_BSTAMP(i) do { __builtin_amdgcn_s_waitcnt(0); bstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define NBK_BSTAMP(i) do { } while (0)
#endif
#ifndef NBK_BF32_WAVES
#define NBK_BF32_WAVES 5
#endif
#ifndef NBK_ZMASK
#define NBK_ZMASK 4u               // bit p: the z coordinates of slots 2p, 2p + 1 live in LDS, not in registers (see cz in the kernel)
#define NBK_ZFIRST 4               // first such slot and how many slots from there the LDS area backs
#define NBK_ZSLOTS 2
#endif
template <int S, bool WH>      // WH: the world holds hulls (scenes without them get a kernel without that branch: primitive scenes lost 8 us to it)
__global__ __launch_bounds__(64, NBK_BF32_WAVES) void k_broad_f32(DevModel m, EdgeSrc es, const double* __restrict__ q, int64_t B, double thr,
                                                   uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
                                                   unsigned long long* __restrict__ q_count, unsigned long long* __restrict__ q_items,
                                                   unsigned long long cap, const float* __restrict__ tab) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nq = m.n_q;
    const int W = m.n_wshapes;
    double* lds_raw = lds;
    const int qrows = f32_qrows(nq, S);
    const int qcap = qrows * (WAVE * 2);                                          // queue entries the slab holds
    float* lds_fr = reinterpret_cast<float*>(lds_raw + WAVE * qrows);             // saved frames [12*slots][64] float
    float* lds_zp = lds_fr + WAVE * 12 * m.frame_slots + 4 - NBK_ZFIRST * WAVE;    // z of the slots of NBK_ZMASK, indexed by slot: [NBK_ZFIRST .. NBK_ZFIRST + NBK_ZSLOTS) are backed (see cz below)
    unsigned* lds_queue = reinterpret_cast<unsigned*>(lds_raw);
    // launch-uniform tables (k_prepare_f32): scalar loads
    const FTab ft = ftab_view(tab, W);
    const float* tab_rkey = ft.rkey; const float* tab_rcert = ft.rcert;
    const float* tab_wkey = ft.wkey; const float* tab_wtc = ft.wtc; const int* tab_wp = ft.wp;
    const float* tab_wcert = ft.wcert; const float* tab_wcin = ft.wcin; const float* tab_rho = ft.rho;
    const float up = 1.0f + 2.4e-7f;
    const int64_t Beff = effective_batch(es, B);
    if (base >= Beff) return;               // edge mode: the launch covers the scratch's capacity, this block lies beyond the samples
#ifdef NBK_BF32_STAMP
    unsigned long long bstamp[6] = {0, 0, 0, 0, 0, 0};
#endif
    NBK_BSTAMP(0);
    const int rows_i = (int)((Beff - base) < WAVE ? (Beff - base) : WAVE);
    if (es.map != nullptr) {
        if (lane < rows_i) {
            unsigned e;
            const double t = edge_t(es, es.map[base + lane], e);
            const double omt = 1.0 - t;
            const double* sp = es.starts + (size_t)e * nq;
            const double* gp = es.goals + (size_t)e * nq;
            for (int j = 0; j < nq; ++j) { const double a = omt * sp[j]; const double bb = t * gp[j]; lds_raw[lane * nq + j] = a + bb; }
        } else {
            for (int j = 0; j < nq; ++j) lds_raw[lane * nq + j] = 0.0;
        }
    } else {
        const int total = rows_i * nq;
        const double* src = q + base * nq;
#if defined(NBK_BF32_ABL) && NBK_BF32_ABL == 3      // timing experiment: no global read of q
        if (true) { for (int i = lane; i < WAVE * nq; i += WAVE) lds_raw[i] = 0.001 * (double)(i + (int)blockIdx.x); } else
#endif
        if (rows_i == WAVE && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (total % 2 == 0)) {
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(lds_raw);
            for (int i = lane; i < total / 2; i += WAVE) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += WAVE) lds_raw[i] = src[i];
            for (int i = total + lane; i < WAVE * nq; i += WAVE) lds_raw[i] = 0.0;
        }
    }
    __syncthreads();
    const bool active = lane < rows_i;
    NBK_BSTAMP(1);
    bool hit = row_nonfinite(lds_raw + lane * nq, nq, 1);        // non-finite joint values: colliding, nothing queued
    // ---- sweep in float32: centres into registers ---------------------------------------------------------------------------
    // (one register tuple per coordinate: consecutive shapes are aligned register pairs, so the robot-robot slots below run two at a
    // time on v_pk_add_f32 / v_pk_fma_f32, and the chain sweep writes centre `s` with a run-time s through VGPR indexing)
    float cxa[S], cya[S], cza[S];
#define cx(i_) cxa[i_]
#define cy(i_) cya[i_]
    // (the 36 centre floats of 12 slots do not all fit next to the row arithmetic at 5 waves per SIMD: the compiler spilled one z pair
    // -- 8 MB of scratch writes per 1e6 configurations --; the z of slots 4 and 5 stay in LDS instead and are read where they are used)
#define NBK_ZL(i_) (((NBK_ZMASK >> ((i_) >> 1)) & 1u) != 0u)
#define cz(i_) (NBK_ZL(i_) ? lds_zp[(i_) * WAVE + lane] : cza[i_])
#define cx2v(i_) V2f{cxa[2 * (i_)], cxa[2 * (i_) + 1]}
#define cy2v(i_) V2f{cya[2 * (i_)], cya[2 * (i_) + 1]}
#define cz2v(i_) (NBK_ZL(2 * (i_)) ? V2f{lds_zp[(2 * (i_)) * WAVE + lane], lds_zp[(2 * (i_) + 1) * WAVE + lane]} : V2f{cza[2 * (i_)], cza[2 * (i_) + 1]})
    float rmax = m.f_reach, qabs = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) { cxa[i] = 0.0f; cya[i] = 0.0f; cza[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < S; ++i) if (NBK_ZL(i)) lds_zp[i * WAVE + lane] = 0.0f;
    if (m.f_chain) {
        // ---- serial chains of at most 8 joints (every arm): the joint loop is unrolled at COMPILE time, so every per-joint table sits
        // at a static offset (scalar loads the compiler issues early, no dependent load -> wait -> branch chains, no address
        // arithmetic), the q values of all joints are read up front, and the shapes of a frame are a run-time loop that writes
        // centre `s` through VGPR indexing (s_set_gpr_idx_on); the shapes' local offsets sit in the lanes of one register and are
        // broadcast with v_readlane.  The loop-per-shape form below spent half its instructions on scalar bookkeeping.
        const float* ftb = m.f_tab;
        const int nshape3 = 3 * m.n_rshapes;
        const int tlv = lane < nshape3 ? __builtin_bit_cast(int, ftb[m.f_tl + lane]) : 0;
        const int* jb = m.joint_shape_begin;
        const int J = m.n_joints;
        // joint kind | q column << 8 of all eight joint slots in one scalar load (slots beyond the robot's joints hold 0), then the
        // eight q values in eight independent LDS reads -- no load -> wait -> read -> wait chain per joint
        struct alignas(32) Meta8 { unsigned v[8]; };
        const Meta8 mt = *reinterpret_cast<const Meta8*>(ftb + m.f_meta);
        double qd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qd[k] = lds_raw[lane * nq + (int)(mt.v[k] >> 8)];
        float qv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { qv[k] = k < J ? (float)qd[k] : 0.0f; qabs += __builtin_fabsf(qv[k]); }
        // the z coordinates travel through the (by now dead) q slab: the compiler keeps only two of the three centre arrays in
        // registers under run-time indexing, the third would live in scratch for the rest of the kernel
        __syncthreads();
        float* lds_cz = reinterpret_cast<float*>(lds_raw);
        XfP T;
        {
            const float* bp = ftb + m.f_base;
#pragma unroll
            for (int k = 0; k < 3; ++k) { T.Rc[k] = V2f{bp[k], bp[4 + k]}; T.R2[k] = bp[8 + k]; }
            T.tc = V2f{bp[3], bp[7]}; T.t2 = bp[11];
        }
#define NBK_CHAIN_SHAPES(s0_, s1_)                                                                                              \
        for (int sh_ = (s0_); sh_ < (s1_); ++sh_) {                                                                             \
            const float tl0_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tlv, 3 * sh_));                              \
            const float tl1_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tlv, 3 * sh_ + 1));                          \
            const float tl2_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(tlv, 3 * sh_ + 2));                          \
            const V2f cp_ = fma2(T.Rc[2], splat2(tl2_), fma2(T.Rc[1], splat2(tl1_), fma2(T.Rc[0], splat2(tl0_), T.tc)));        \
            const float c2_ = __builtin_fmaf(T.R2[2], tl2_, __builtin_fmaf(T.R2[1], tl1_, __builtin_fmaf(T.R2[0], tl0_, T.t2))); \
            cxa[sh_] = cp_.x; cya[sh_] = cp_.y; lds_cz[sh_ * WAVE + lane] = c2_;                                                 \
            rmax = __builtin_fmaxf(rmax, __builtin_fmaxf(__builtin_fabsf(cp_.x), __builtin_fmaxf(__builtin_fabsf(cp_.y), __builtin_fabsf(c2_)))); \
        }
        NBK_CHAIN_SHAPES(jb[0], jb[1])
        // the constants of joint k + 1 are loaded (unconditionally: the table has a spare entry) before joint k is applied
        JPk jcur = *reinterpret_cast<const JPk*>(ftb + m.f_pk);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const JPk jnxt = *reinterpret_cast<const JPk*>(ftb + m.f_pk + 20 * (k + 1 < 8 ? k + 1 : 7));
            if (k < J) {
                joint_apply_p(m, k, (int)(mt.v[k] & 255u), jcur, T, qv[k], T);
                NBK_CHAIN_SHAPES(jb[k + 1], jb[k + 2])
            }
            jcur = jnxt;
        }
#undef NBK_CHAIN_SHAPES
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float zi = i < m.n_rshapes ? lds_cz[i * WAVE + lane] : 0.0f;
            if (NBK_ZL(i)) lds_zp[i * WAVE + lane] = zi; else cza[i] = zi;
        }
    } else {
        XfF bpose;
        const float* bp = m.f_tab + m.f_base;
#pragma unroll
        for (int i = 0; i < 3; ++i) { bpose.R[3 * i] = bp[4 * i]; bpose.R[3 * i + 1] = bp[4 * i + 1]; bpose.R[3 * i + 2] = bp[4 * i + 2]; bpose.t[i] = bp[4 * i + 3]; }
        XfF T = bpose;
        int kcur = -1;
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) {
            if (sidx < m.n_rshapes) {
                const int f = m.rs_frame[sidx];
                while (kcur < f) {
                    ++kcur;
                    const int k = kcur;
                    const int ld = m.joint_load[k];
                    XfF Pf;
                    if (ld == -2) Pf = T;
                    else if (ld == -1) Pf = bpose;
                    else {
#pragma unroll
                        for (int e = 0; e < 9; ++e) Pf.R[e] = lds_fr[(ld * 12 + e) * WAVE + lane];
#pragma unroll
                        for (int e = 0; e < 3; ++e) Pf.t[e] = lds_fr[(ld * 12 + 9 + e) * WAVE + lane];
                    }
                    const float qk = (float)lds_raw[lane * nq + m.joint_qidx[k]];
                    qabs += __builtin_fabsf(qk);
                    joint_apply_f(m, k, Pf, qk, T);
                    const int sv = m.joint_save[k];
                    if (sv >= 0) {
#pragma unroll
                        for (int e = 0; e < 9; ++e) lds_fr[(sv * 12 + e) * WAVE + lane] = T.R[e];
#pragma unroll
                        for (int e = 0; e < 3; ++e) lds_fr[(sv * 12 + 9 + e) * WAVE + lane] = T.t[e];
                    }
                }
                const float* tl = m.f_tab + m.f_tl + 3 * sidx;
                const XfF& F = f < 0 ? bpose : T;
                float c[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) c[i] = __builtin_fmaf(F.R[3 * i + 2], tl[2], __builtin_fmaf(F.R[3 * i + 1], tl[1], __builtin_fmaf(F.R[3 * i], tl[0], F.t[i])));
                cx(sidx) = c[0]; cy(sidx) = c[1]; cz(sidx) = c[2];
                rmax = __builtin_fmaxf(rmax, __builtin_fmaxf(__builtin_fabsf(c[0]), __builtin_fmaxf(__builtin_fabsf(c[1]), __builtin_fabsf(c[2]))));
            }
        }
    }
    // both centres are off by at most e = rmax * (f_eps + the angle error of the float32 joint values, 6e-8 |q| each, x4)
    const float e2 = 2.0f * rmax * __builtin_fmaf(2.4e-7f, qabs, m.f_eps);
    __syncthreads();            // the q slab is dead from here on: its LDS region becomes the item queue
    // ---- certified hits, fused into the candidate tests at one compare per slot: the balls inscribed in two shapes (radius:
    // margin + smallest half extent of the core) overlap by more than the slack => the pair collides in float64 as well, whatever
    // the narrowphase would compute in more digits.  The thresholds (tab_?cert2) are squared and carry the STATIC slack bound
    // f_e2max, so they are scalars; a lane whose own slack is larger (prismatic travel, huge joint values) never certifies.
    // A configuration known to collide queues nothing: the world shapes go first, then the robot rows -- a hit found in a row
    // stops that row and the later ones (the earlier rows' items stay: 0.097 instead of 0.088 items per configuration).
    int qn = 0;
    const int n_reach = *ft.n_reach;
#if defined(NBK_BF32_ABL) && (NBK_BF32_ABL == 1 || NBK_BF32_ABL == 3)      // timing experiment (tools/build_variant.sh): sweep only, no pair stage
    {
        float acc = e2;
#pragma unroll
        for (int i = 0; i < S; ++i) acc += cx(i) + cy(i) + cz(i);
        const unsigned long long word = __builtin_amdgcn_ballot_w64((hit || acc == 12345.0f) && active);
        if (mask_bits != nullptr && lane == 0) mask_bits[blockIdx.x] = word;
        return;
    }
#endif
    // ---- fast pair stage: every lane of the wave is within the static slack bound (always, for revolute robots with |q| sums below
    // 64 rad), so every threshold is a scalar that k_prepare_f32 has squared already: a slot costs the squared centre distance and
    // two compares.  The compare results stay lane masks in scalar registers -- no per-lane survivor bits -- and are queued slot by
    // slot (a pair at a time) after the world shape's / the row's certified hits are known.
    NBK_BSTAMP(2);
    if (__builtin_amdgcn_ballot_w64(!(e2 <= m.f_e2max)) == 0ull) {
        const float* tab_rneg = ft.rneg;
        const float* tab_rnd = ft.rnd;
        const float* tab_wkey2 = ft.wkey2;
        // one queue append: lanes with `cond_` get consecutive slots behind the pending items.  No "nearly full" test here:
        // NBK_ROOM makes room for a whole row of slots before the row's appends start
#define NBK_ENQUEUE(cond_, pidx_)                                                                                               \
        {                                                                                                                       \
            const bool c_ = (cond_);                                                                                            \
            const unsigned long long cm_ = __builtin_amdgcn_ballot_w64(c_);                                                     \
            if (cm_ != 0ull) {                                                                                                  \
                if (c_) {                                                                                                       \
                    const int pos_ = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm_, 0u)); \
                    lds_queue[pos_] = ((unsigned)(pidx_) << 6) | (unsigned)lane;                                                \
                }                                                                                                               \
                qn += __builtin_popcountll(cm_);                                                                                \
            }                                                                                                                   \
        }
#define NBK_ROOM(slots_)                                                                                                        \
        if (qn > qcap - (slots_) * WAVE) { flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf); qn = 0; }
#if defined(NBK_BF32_ABL) && NBK_BF32_ABL == 5
        for (int wi = 0; wi < (e2 == 12345.0f ? n_reach : 0); ++wi) {
#else
        for (int wi = 0; wi < n_reach; ++wi) {
#endif
            const int w = ft.wlist[wi];
            const float* wc = m.f_tab + m.f_wc + 18 * w;
            const int wk = m.ws_kind[w];
            // this world shape's rows of the slot tables: one 64-byte scalar load each
            const Row16f wkey2r = *reinterpret_cast<const Row16f*>(tab_wkey2 + w * 16);
            const Row16f wcertr = *reinterpret_cast<const Row16f*>(tab_wcert + w * 16);
            const Row16f wkeyr = *reinterpret_cast<const Row16f*>(tab_wkey + w * 16);
            bool c[S];
            bool ch = false;
#pragma unroll
            for (int a = 0; a < S; ++a) c[a] = false;
            if (wk == K_PLANE) {
#pragma unroll
                for (int a = 0; a < S; ++a) {
                    if (a < m.n_rshapes && tab_wp[w * 16 + a] >= 0) {
                        const float dx = cx(a) - wc[0], dy = cy(a) - wc[1], dz = cz(a) - wc[2];
                        const float hc = __builtin_fmaf(dz, wc[11], __builtin_fmaf(dy, wc[10], dx * wc[9]));
                        c[a] = hc < wkey2r.v[a];
                        ch = ch || (hc < wcertr.v[a]);
                    }
                }
            } else if (wk == K_BOX) {
                // Two robot shapes (2i, 2i + 1) per trip against the box, no branch and no lane mask per slot: the centre's squared distance
                // dd to the box centre and ex2 to the (core) box itself -- ex_j = |d . axis_j| - h_j, clamped at 0, squared and summed --
                // and mx = max_j ex_j (< 0: the centre is inside).  Every verdict is the sign of a difference with a per-slot scalar
                // (k_prepare_f32: all thresholds carry the STATIC slack bound, which this stage's lanes stay within):
                //   candidate  dd < wkey2 (bounding spheres) and ex2 < cull2 (closer to the box than tc+ + rho + slack)
                //   certain hit, outside  candidate, ex2 > 0 and ex2 < cin (inside the ball inscribed in the shape, slack taken off)
                //   certain hit, inside   candidate, mx < 0, mx < -g (deeper than -tc + slack) and dd < kin
                const float* wb = ft.wbx + w * 96;
                int cwv[S];
                int acc_c = 0, acc_h = 0;
#pragma unroll
                for (int i = 0; i < S / 2; ++i) {
                    const V2f dx = cx2v(i) - splat2(wc[0]), dy = cy2v(i) - splat2(wc[1]), dz = cz2v(i) - splat2(wc[2]);
                    const V2f dd = fma2(dz, dz, fma2(dy, dy, dx * dx));
                    V2f ex[3], ex2 = V2f{0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const V2f pj = fma2(dz, splat2(wc[5 + 3 * j]), fma2(dy, splat2(wc[4 + 3 * j]), dx * splat2(wc[3 + 3 * j])));
                        ex[j] = V2f{__builtin_fabsf(pj.x) - wc[12 + j], __builtin_fabsf(pj.y) - wc[12 + j]};
                        const V2f cl = V2f{__builtin_fmaxf(ex[j].x, 0.0f), __builtin_fmaxf(ex[j].y, 0.0f)};
                        ex2 = fma2(cl, cl, ex2);
                    }
                    const V2f mx = V2f{__builtin_fmaxf(ex[0].x, __builtin_fmaxf(ex[1].x, ex[2].x)), __builtin_fmaxf(ex[0].y, __builtin_fmaxf(ex[1].y, ex[2].y))};
                    const float* tb = wb + 12 * i;
                    const V2i s1 = __builtin_bit_cast(V2i, dd - V2f{tb[0], tb[1]});
                    const V2i s2 = __builtin_bit_cast(V2i, ex2 - V2f{tb[2], tb[3]});
                    const V2i s3 = __builtin_bit_cast(V2i, ex2 - V2f{tb[4], tb[5]});
                    const V2i s4 = __builtin_bit_cast(V2i, dd - V2f{tb[6], tb[7]});
                    const V2i s5 = __builtin_bit_cast(V2i, mx + V2f{tb[8], tb[9]});
                    const V2i nz = __builtin_bit_cast(V2i, V2f{0.0f, 0.0f} - ex2);         // sign set <=> ex2 > 0 (a true subtraction: +0 - +0 = +0)
                    const V2i mi = __builtin_bit_cast(V2i, mx);
                    const V2i cand = s1 & s2;
                    const V2i certh = cand & ((s3 & nz) | (mi & s5 & s4));
                    cwv[2 * i] = cand.x; cwv[2 * i + 1] = cand.y;
                    acc_c |= cand.x | cand.y;
                    acc_h |= certh.x | certh.y;
                }
                hit = hit || (acc_h < 0);
                const bool live = active && !hit && !(NBK_DBG(m) & 4);
                if (__builtin_amdgcn_ballot_w64(acc_c < 0 && live) != 0ull) {
                    NBK_ROOM(S)
#pragma unroll
                    for (int a = 0; a < S; ++a)
                        if (a < m.n_rshapes) NBK_ENQUEUE(cwv[a] < 0 && live, tab_wp[w * 16 + a]);
                }
                continue;
            } else if (WH && wk == K_HULL) {
                // the box slot's form with the hull's local bounding box: candidate = bounding spheres (dd < wkey2) and the centre closer
                // to that box than tc+ + rho + slack (a cull only: the box contains the hull); certain hit = the inscribed balls (dd < cert)
                const float* ob = m.f_tab + m.f_wobb + 6 * w;
                const float* wb = ft.wbx + w * 96;
                int cwv[S];
                int acc_c = 0, acc_h = 0;
#pragma unroll
                for (int i = 0; i < S / 2; ++i) {
                    const V2f dx = cx2v(i) - splat2(wc[0]), dy = cy2v(i) - splat2(wc[1]), dz = cz2v(i) - splat2(wc[2]);
                    const V2f dd = fma2(dz, dz, fma2(dy, dy, dx * dx));
                    V2f ex2 = V2f{0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const V2f pj = fma2(dz, splat2(wc[5 + 3 * j]), fma2(dy, splat2(wc[4 + 3 * j]), dx * splat2(wc[3 + 3 * j]))) - splat2(ob[j]);
                        const V2f cl = V2f{__builtin_fmaxf(__builtin_fabsf(pj.x) - ob[3 + j], 0.0f), __builtin_fmaxf(__builtin_fabsf(pj.y) - ob[3 + j], 0.0f)};
                        ex2 = fma2(cl, cl, ex2);
                    }
                    const float* tb = wb + 12 * i;
                    const V2i s1 = __builtin_bit_cast(V2i, dd - V2f{tb[0], tb[1]});
                    const V2i s2 = __builtin_bit_cast(V2i, ex2 - V2f{tb[2], tb[3]});
                    const V2i s3 = __builtin_bit_cast(V2i, dd - V2f{tb[4], tb[5]});
                    const V2i cand = s1 & s2;
                    cwv[2 * i] = cand.x; cwv[2 * i + 1] = cand.y;
                    acc_c |= cand.x | cand.y;
                    acc_h |= s3.x | s3.y;
                }
                hit = hit || (acc_h < 0);
                const bool live = active && !hit && !(NBK_DBG(m) & 4);
                if (__builtin_amdgcn_ballot_w64(acc_c < 0 && live) != 0ull) {
                    NBK_ROOM(S)
#pragma unroll
                    for (int a = 0; a < S; ++a)
                        if (a < m.n_rshapes) NBK_ENQUEUE(cwv[a] < 0 && live, tab_wp[w * 16 + a]);
                }
                continue;
            } else {
#pragma unroll
                for (int a = 0; a < S; ++a) {
                    if (a < m.n_rshapes && wkeyr.v[a] >= 0.0f) {
                        const float dx = cx(a) - wc[0], dy = cy(a) - wc[1], dz = cz(a) - wc[2];
                        const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                        c[a] = dd < wkey2r.v[a];
                        ch = ch || (dd < wcertr.v[a]);
                    }
                }
            }
            hit = hit || ch;
            const bool live = active && !hit && !(NBK_DBG(m) & 4);
            bool anyc = false;
#pragma unroll
            for (int a = 0; a < S; ++a) anyc = anyc || c[a];
            if (__builtin_amdgcn_ballot_w64(anyc && live) != 0ull) {          // one branch per world shape; most have no candidate
                NBK_ROOM(S)
#pragma unroll
                for (int a = 0; a < S; ++a)
                    if (a < m.n_rshapes) NBK_ENQUEUE(c[a] && live, tab_wp[w * 16 + a]);
            }
        }
        NBK_BSTAMP(3);
        if (m.bq_count[1] > 0) {
#pragma unroll
            for (int a = 0; a < S - 1; ++a) {
                // row a of the slot table: one 64-byte scalar load.  Two slots (b, b + 1) per trip: e = |ca - cb|^2 - k2 as an fma
                // chain that starts from -k2; a candidate is a NEGATIVE e, so the row's verdict is the sign bit of one OR-accumulator --
                // no compare, no lane mask, no scalar instruction per slot.  Slots b <= a of the first pair and slots that are not
                // pairs hold +1: never negative.  Only a row with a candidate (one branch per row) looks further: f = e + (k2 - cert)
                // is negative for a certified hit (a certified pair is a candidate: cert < k2), then the candidates of the lanes that
                // are still undecided are queued from the e's kept in registers.
                const Row16f nk = *reinterpret_cast<const Row16f*>(tab_rneg + a * 16);
                const V2f ax2 = V2f{cx(a), cx(a)}, ay2 = V2f{cy(a), cy(a)}, az2 = V2f{cz(a), cz(a)};
                V2f ev[S / 2];
                int acc_e = 0;
#pragma unroll
                for (int i = (a + 1) / 2; i < S / 2; ++i) {
                    ev[i] = slot_e2(ax2 - cx2v(i), ay2 - cy2v(i), az2 - cz2v(i), V2f{nk.v[2 * i], nk.v[2 * i + 1]});
                    // (bit-cast the PAIR, then OR its halves: `bit_cast<int>(e.x) | bit_cast<int>(e.y)` under a sign test loses the
                    // second element in this compiler -- ROCm 7.2 clang folds it to element 0 already in the IR)
                    const V2i ei = __builtin_bit_cast(V2i, ev[i]);
                    acc_e |= ei.x | ei.y;
                }
#if defined(NBK_BF32_ABL) && NBK_BF32_ABL == 4
                if (__builtin_amdgcn_ballot_w64(acc_e < 0 && active && !hit && e2 == 12345.0f) != 0ull) {
#else
                if (__builtin_amdgcn_ballot_w64(acc_e < 0 && active && !hit) != 0ull) {
#endif
                    const Row16f nd = *reinterpret_cast<const Row16f*>(tab_rnd + a * 16);
                    int acc_f = 0;
#pragma unroll
                    for (int i = (a + 1) / 2; i < S / 2; ++i) {
                        const V2i fi = __builtin_bit_cast(V2i, ev[i] + V2f{nd.v[2 * i], nd.v[2 * i + 1]});
                        acc_f |= fi.x | fi.y;
                    }
                    hit = hit || (acc_f < 0);
                    const bool live = active && !hit;
                    NBK_ROOM(S)
#pragma unroll
                    for (int b = a + 1; b < S; ++b) {
                        const V2i ei = __builtin_bit_cast(V2i, ev[b / 2]);
                        NBK_ENQUEUE(((b % 2) ? ei.y : ei.x) < 0 && live, ft.rp[a * 16 + b]);
                    }
                }
            }
        }
#undef NBK_ENQUEUE
#undef NBK_ROOM
        NBK_BSTAMP(4);
        if (qn > 0) flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf);
        const unsigned long long word = __builtin_amdgcn_ballot_w64(hit && active);
        if (mask_bits != nullptr && lane == 0) mask_bits[blockIdx.x] = word;
        if (mask_bytes != nullptr && active) mask_bytes[base + lane] = hit ? 1 : 0;
#ifdef NBK_BF32_STAMP
        NBK_BSTAMP(5);
        if (lane == 0) { unsigned long long* pr_ = g_broad_prof + 8 * (blockIdx.x & 16383u); for (int e_ = 0; e_ < 5; ++e_) pr_[e_] += bstamp[e_ + 1] - bstamp[e_]; pr_[7] += 1ull; }
#endif
        return;
    }
    // ---- general pair stage (some lane's slack exceeds the static bound: prismatic travel, huge joint values) ----------------------
    const bool cert_ok = e2 <= m.f_e2max;
    bool certh = false;
    for (int wi = 0; wi < n_reach; ++wi) {
        const int w = ft.wlist[wi];                 // only the world shapes within somebody's reach at this threshold
        const float* wc = m.f_tab + m.f_wc + 18 * w;
        const int wk = m.ws_kind[w];
        unsigned long long bits = 0ull;
        if (wk == K_PLANE) {
#pragma unroll
            for (int a = 0; a < S; ++a) {
                if (a < m.n_rshapes && tab_wp[w * 16 + a] >= 0) {          // uniform: not a pair, or out of reach for good
                    const float key = tab_wkey[w * 16 + a];
                    const float rhoA = tab_rho[a];
                    const float dx = cx(a) - wc[0], dy = cy(a) - wc[1], dz = cz(a) - wc[2];
                    const float hc = __builtin_fmaf(dz, wc[11], __builtin_fmaf(dy, wc[10], dx * wc[9]));
                    const bool cand = (tab_wp[w * 16 + a] >= 0) && !((hc - rhoA) >= key + e2);
                    bits |= cand ? (1ull << a) : 0ull;
                    certh = certh || (hc < tab_wcert[w * 16 + a]);           // the inscribed ball dips below the plane
                }
            }
        } else if (wk == K_BOX) {
#pragma unroll
            for (int a = 0; a < S; ++a) {
                if (a < m.n_rshapes && tab_wkey[w * 16 + a] >= 0.0f) {
                    const float rs = tab_wkey[w * 16 + a];
                    const float tc = tab_wtc[w * 16 + a];
                    const float rho = tab_rho[a];
                    const float dx = cx(a) - wc[0], dy = cy(a) - wc[1], dz = cz(a) - wc[2];
                    const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    const float r = rs + e2;
                    bool cand = rs >= 0.0f && dd < r * r * up;
                    // midphase on the exact box (distance of the centre outside, depth inside) -- only when some lane of
                    // the wave passed the sphere test: far (shape, box) combinations cost six instructions
                    if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
                        float ex2 = 0.0f, g = 3.4e38f;
                        bool inside = true;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float axj = __builtin_fabsf(__builtin_fmaf(dz, wc[5 + 3 * j], __builtin_fmaf(dy, wc[4 + 3 * j], dx * wc[3 + 3 * j])));
                            const float exj = axj - wc[12 + j];
                            if (exj > 0.0f) { inside = false; ex2 = __builtin_fmaf(exj, exj, ex2); }
                            g = __builtin_fminf(g, wc[12 + j] - axj);
                        }
                        if (!inside) {
                            // outside by more than the slack in every float64 reading: free when far enough (tc >= 0 only)
                            const float rr = ((tc > 0.0f ? tc : 0.0f) + rho) + e2;       // tc < 0: disjoint is enough (device-only cull)
                            if (ex2 >= rr * rr * up) cand = false;
                            // the centre is closer to the box than the radius of the ball inscribed in the shape: certain hit
                            if (cand && ex2 < tab_wcin[w * 16 + a]) certh = true;
                        } else if (cand && g > -tc + e2 && rs > e2 && dd * up < (rs - e2) * (rs - e2)) {
                            hit = true;         // inside deeper than -tc, and inside the sphere test, in float64 as well: certain hit
                        }
                    }
                    bits |= cand ? (1ull << a) : 0ull;
                }
            }
        } else {
            const bool is_hull = wk == K_HULL;
            const float* ob = m.f_tab + m.f_wobb + 6 * w;                // hulls: local bounding box (a cull, as in the fast stage)
#pragma unroll
            for (int a = 0; a < S; ++a) {
                if (a < m.n_rshapes && tab_wkey[w * 16 + a] >= 0.0f) {
                    const float rs = tab_wkey[w * 16 + a];
                    const float r = rs + e2;
                    const float dx = cx(a) - wc[0], dy = cy(a) - wc[1], dz = cz(a) - wc[2];
                    const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    bool cand = rs >= 0.0f && dd < r * r * up;
                    if (is_hull && __builtin_amdgcn_ballot_w64(cand) != 0ull) {
                        const float tc = tab_wtc[w * 16 + a], rho = tab_rho[a];
                        float ex2 = 0.0f;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float xj = __builtin_fmaf(dz, wc[5 + 3 * j], __builtin_fmaf(dy, wc[4 + 3 * j], dx * wc[3 + 3 * j])) - ob[j];
                            const float exj = __builtin_fabsf(xj) - ob[3 + j];
                            if (exj > 0.0f) ex2 = __builtin_fmaf(exj, exj, ex2);
                        }
                        const float rr = ((tc > 0.0f ? tc : 0.0f) + rho) + e2;
                        if (ex2 >= rr * rr * up) cand = false;
                    }
                    bits |= cand ? (1ull << a) : 0ull;
                    certh = certh || (dd < tab_wcert[w * 16 + a]);           // inscribed balls overlap
                }
            }
        }
        hit = hit || (cert_ok && certh);
        if (!active || hit || (NBK_DBG(m) & 4)) bits = 0ull;
        while (true) {
            const bool has = bits != 0ull;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
            if (bal == 0ull) break;
            if (has) {
                const int bit = __builtin_ctzll(bits);
                bits &= bits - 1ull;
                const unsigned p = (unsigned)tab_wp[w * 16 + bit];
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                lds_queue[pos] = (p << 6) | (unsigned)lane;
            }
            qn += __builtin_popcountll(bal);
            if (qn > BQ_CAP - WAVE) { flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf); qn = 0; }
        }
    }
    if (m.bq_count[1] > 0) {
#pragma unroll
        for (int a = 0; a < S - 1; ++a) {
            unsigned long long bits = 0ull;
#pragma unroll
            for (int b = a + 1; b < S; ++b) {
                const float rs = tab_rkey[a * 16 + b];
                const float r = rs + e2;
                const float dx = cx(a) - cx(b), dy = cy(a) - cy(b), dz = cz(a) - cz(b);
                const float dd = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                bits |= (rs >= 0.0f && dd < r * r * up) ? (1ull << b) : 0ull;
                certh = certh || (dd < tab_rcert[a * 16 + b]);
            }
            hit = hit || (cert_ok && certh);
            if (!active || hit) bits = 0ull;
            while (true) {
                const bool has = bits != 0ull;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
                if (bal == 0ull) break;
                if (has) {
                    const int bit = __builtin_ctzll(bits);
                    bits &= bits - 1ull;
                    const unsigned p = (unsigned)ft.rp[a * 16 + bit];
                    const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    lds_queue[pos] = (p << 6) | (unsigned)lane;
                }
                qn += __builtin_popcountll(bal);
                if (qn > BQ_CAP - WAVE) { flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf); qn = 0; }
            }
        }
    }
    if (qn > 0) flush_items(m, lds_queue, qn, base, q_count, q_items, cap, lane, es.ovf);
    const unsigned long long word = __builtin_amdgcn_ballot_w64(hit && active);
    if (mask_bits != nullptr && lane == 0) mask_bits[blockIdx.x] = word;
    if (mask_bytes != nullptr && active) mask_bytes[base + lane] = hit ? 1 : 0;
}
#undef cx
#undef cy
#undef cz
#undef cx2v
#undef cy2v
#undef cz2v

// core of shape `ref` (robot: from the replayed frame T; world: table).  Everything here is per lane.
NBK_DEV void build_core(const DevModel& m, int ref, const Xf& T, Core& o) {
    // the constants of either table are read through one pointer and stored once, outside the branch: with the stores inside,
    // the compiler sank "the last store of either branch" into one store at a lane-varying address (o.rho on one side, o.c[0]
    // on the other), which pinned those members of both cores in scratch -- and every later read of them
    const bool robot = ref >= 0;
    const int w = robot ? 0 : ~ref;
    const double* wc = m.ws_core + 18 * w;
    const double* cc = robot ? m.rs_core + 6 * ref : wc + 12;
    o.kind = robot ? m.rs_kind[ref] : m.ws_kind[w];
    o.h[0] = cc[0]; o.h[1] = cc[1]; o.h[2] = cc[2]; o.rad = cc[3]; o.margin = cc[4]; o.rho = cc[5];
    if (robot) {
        const double* loc = m.rs_local + 12 * ref;
        double Rl[9], tl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { Rl[3 * i] = loc[4 * i]; Rl[3 * i + 1] = loc[4 * i + 1]; Rl[3 * i + 2] = loc[4 * i + 2]; tl[i] = loc[4 * i + 3]; }
        xf_mul_pos(T, tl, o.c);
        if (o.kind == K_SEG || o.kind == K_CYL) {
            xf_mul_col(T, Rl, 2, o.ax[2]);
        } else if (o.kind == K_BOX || o.kind == K_HULL) {
#pragma unroll
            for (int j = 0; j < 3; ++j) xf_mul_col(T, Rl, j, o.ax[j]);
        }
    } else {
        o.c[0] = wc[0]; o.c[1] = wc[1]; o.c[2] = wc[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { o.ax[j][0] = wc[3 + 3 * j]; o.ax[j][1] = wc[4 + 3 * j]; o.ax[j][2] = wc[5 + 3 * j]; }
    }
}

// k_narrow: one 64-lane workgroup takes 64 queue items at a time.
//   phase 1 (one item per lane, dense): FK replay of the item's two primitives, cores, the float64 bounding-sphere
//            step, planes / box midphase / closed forms;
//   phase 2: the lanes whose item is still undecided walk GJK, one iteration per trip of the wave, until the last one
//            has its verdict (iteration counts differ: mean 4, max ~10 per chunk).  An LDS pool that handed finished
//            lanes new items existed while chunks were larger than the wave; with 64-item chunks it never had
//            anything left to hand out and only cost 18 KB of LDS per workgroup.
constexpr int NARROW_T = 64;
constexpr int HULL_LDS_MAX = 16 * 1024;   // k_narrow*: the scene's hull vertices live in LDS when they fit this (the per-lane vertex loop of
                                          // the hull support is 3 loads per vertex at lane-varying addresses: LDS serves those several
                                          // times faster than the vector memory path)
static inline size_t narrow_hull_lds(int hull_blob_n) { return (hull_blob_n > 0 && (size_t)hull_blob_n * 8 <= (size_t)HULL_LDS_MAX) ? (size_t)hull_blob_n * 8 : 0; }
#ifndef NARROW_WAVES_GEN
#define NARROW_WAVES_GEN 2
#endif
constexpr int NARROW_WAVES_BOOL = 2;     // waves per SIMD the boolean-only narrowphase is compiled for (3 needs ~120 spilled
                                         // registers: measured slower)

NBK_DEV void mark_hit(long long b, uint64_t* mask_bits, uint8_t* mask_bytes) {
    if (mask_bits != nullptr) atomicOr(reinterpret_cast<unsigned long long*>(mask_bits) + (b >> 6), 1ull << (b & 63));
    if (mask_bytes != nullptr) mask_bytes[b] = 1;
}

// BOOL_ONLY: the host has established tc == 0 for every pair that can reach GJK (threshold 0 and no margins on
// box / cylinder / capsule-vs-solid pairs -- the reference's default in_collision(q) call on sharp shapes); only the
// boolean GJK state is kept.
// in-kernel phase timing of k_narrow (NBK_ABLATE bit 128): per working wave, cycles between stamps, summed
__device__ unsigned long long g_narrow_prof[16];
#define NBK_STAMP(i) do { if (prof) { __builtin_amdgcn_s_waitcnt(0); stamp[i] = __builtin_readcyclecounter(); } } while (0)

// MODE 0: any mix of tc (per item: tc >= 0 without a hull core walks the boolean GJK, inflated by tc when tc > 0; tc < 0 and hull
// pairs run the distance predicate), 1: tc == 0 for every pair that can reach GJK and none has a hull core -- boolean walk
// only, 2: tc < 0 everywhere -- distance predicate only, 3: tc >= 0 everywhere (a positive threshold such as IRIS' 1e-6, a
// margin on every solid, or threshold 0 on a scene with meshes) -- the walk, then the distance predicate for hull pairs and for
// what an inflated walk leaves undecided
template <int MODE>
NBK_DEV void narrow_body(const DevModel& m, const EdgeSrc& es, const double* __restrict__ q, double thr,
                         const unsigned long long* __restrict__ q_items, unsigned long long* q_count,
                         unsigned long long cap, uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
                         double* qstage, unsigned long long* __restrict__ count_next) {
    // block (sub, part): every nparts-th 128-item chunk of sub-queue `sub`
    const unsigned sub = blockIdx.x % NSUB;
    const unsigned part = blockIdx.x / NSUB;
    const unsigned nparts = gridDim.x / NSUB;
    // the other counter set is idle during this call: clear it for the next one
    if (count_next != nullptr && part == 0 && threadIdx.x == 0) { count_next[(size_t)sub * CNT_STRIDE] = 0ull; count_next[(size_t)sub * CNT_STRIDE + CNT_TICKET] = 0ull; }
    // agent-scope loads: the queue was written by another kernel (possibly replayed from a hipGraph).
    // The kernel is latency-bound, so dependent global round trips are kept to three: {count, first item} ->
    // {pair record, q row} -> shape constants.  The first chunk's item is loaded before the count is known (the slot
    // is inside the allocated sub-queue either way; its value is used only when the slot is below the count).
    const bool prof = (NBK_DBG(m) & 128) != 0;
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    NBK_STAMP(0);
    q_items += (unsigned long long)sub * cap;
    const unsigned long long i_first = (unsigned long long)part * NARROW_T + threadIdx.x;
    unsigned long long item_first = 0;
    if (i_first < cap) item_first = __hip_atomic_load(q_items + i_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long n = __hip_atomic_load(q_count + sub * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n > cap) n = cap;
    if (NBK_DBG(m) & 1) n = 0;
    // chunks: workgroup `part` starts with chunk `part`; further chunks are handed out by a ticket per sub-queue, so that a
    // workgroup held up by a slow item (a GJK walk of 30+ steps where 5 are usual) takes fewer chunks instead of finishing its fixed
    // share late -- the kernel ends with its slowest workgroup.  The next ticket is drawn before the current chunk is worked on.
    unsigned long long* ticket = q_count + sub * CNT_STRIDE + CNT_TICKET;
    // hull vertices into LDS (workgroups with work only); the cores' HullRef.hv pointers are redirected after build_core
    double* hull_lds = qstage + NARROW_T * m.n_q;
    const bool hull_staged = m.hull_blob_n > 0 && m.hull_blob_n * 8 <= HULL_LDS_MAX;
    if (hull_staged && (unsigned long long)part * NARROW_T < n) {
        for (int i = threadIdx.x; i < m.hull_blob_n; i += NARROW_T) hull_lds[i] = m.hull_blob[i];
        __syncthreads();
    }
    for (unsigned long long chunk = part; chunk * NARROW_T < n; ) {
        const unsigned long long i0 = chunk * NARROW_T;
        unsigned next_ticket = 0u;
        if (threadIdx.x == 0) next_ticket = (unsigned)atomicAdd(ticket, 1ull);
        chunk = nparts;          // + the ticket, added at the end of the body
        // ---- phase 1 -----------------------------------------------------------------------------------------------
        {
            const unsigned long long i = i0 + threadIdx.x;
            const bool live = i < n;
            unsigned long long item = 0;
            if (i == i_first) item = item_first;
            else if (live) item = __hip_atomic_load(q_items + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!live) item = 0;
            const long long b = (long long)(item >> 20);
            const int p = (int)(item & 0xFFFFFull);
            // dead lanes read record 0 / row 0 of this tile (valid memory) and discard it
            NBK_STAMP(1);
            const int4 info = m.vp_info[p];
            const int ra = live ? info.x : -1, rb = live ? info.y : -1;
            const unsigned ma = live ? (unsigned)info.z : 0u, mb = live ? (unsigned)info.w : 0u;
            Xf TA, TB;
            xf_from12(m.base_pose, TA);
            TB = TA;
            // this lane's q row goes to LDS first, eight loads in flight at a time, instead of one global round trip
            // per joint inside the loop below
            double* myq = qstage + threadIdx.x * m.n_q;
            const int nq1 = m.n_q - 1;
            if (es.map != nullptr) {
                unsigned e;
                const double et = edge_t(es, es.map[b], e);
                const double eomt = 1.0 - et;
                const double* sp = es.starts + (size_t)e * m.n_q;
                const double* gp = es.goals + (size_t)e * m.n_q;
                for (int j0 = 0; j0 < m.n_q; j0 += 4) {
                    double sv[4], gv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int j = (j0 + u) < nq1 ? (j0 + u) : nq1; sv[u] = sp[j]; gv[u] = gp[j]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j0 + u <= nq1) { const double a = eomt * sv[u]; const double bb = et * gv[u]; myq[j0 + u] = a + bb; }
                }
            } else {
                const double* qrow = q + b * m.n_q;
                for (int j0 = 0; j0 < m.n_q; j0 += 8) {
                    double qv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int j = (j0 + u) < nq1 ? (j0 + u) : nq1; qv[u] = qrow[j]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u <= nq1) myq[j0 + u] = qv[u];
                }
            }
            NBK_STAMP(2);
            for (int k = 0; k < ((NBK_DBG(m) & 16) ? 0 : m.n_joints); ++k) {
                const bool in_a = (ma >> k) & 1u, in_b = (mb >> k) & 1u;
                if (__builtin_amdgcn_ballot_w64(in_a || in_b) == 0ull) continue;
                if (in_a || in_b) {
                    const double qk = myq[m.joint_qidx[k]];
                    Xf nxt;
                    joint_apply(m, k, in_a ? TA : TB, qk, nxt);     // common ancestors: TA == TB bit for bit
                    if (in_a) TA = nxt;
                    if (in_b) TB = nxt;
                }
            }
            NBK_STAMP(3);
            bool pooled = false;
            Core A, Bc;
            double tc = 0.0;
            A.kind = K_POINT; Bc.kind = K_POINT;
#pragma unroll
            for (int e = 0; e < 3; ++e) { A.c[e] = 0.0; Bc.c[e] = 1.0; A.h[e] = 0.0; Bc.h[e] = 0.0; A.ax[0][e] = A.ax[1][e] = A.ax[2][e] = 0.0; Bc.ax[0][e] = Bc.ax[1][e] = Bc.ax[2][e] = 0.0; }
            A.rad = Bc.rad = A.margin = Bc.margin = A.rho = Bc.rho = 0.0;
            if (live && !(NBK_DBG(m) & 32)) {
                build_core(m, ra, TA, A);
                build_core(m, rb, TB, Bc);
                if (hull_staged) {
                    // HullRef.hv is redirected to the LDS copy (a flat address: the support routine's loop stays the one every kernel
                    // shares); the extent routine, which the overlap-depth estimate calls per face, gets the LDS address proper in `rad`
                    // (unused by hulls) and reads with ds_read.  The same split for the support routine costs the distance kernels,
                    // which never stage, 5 % (a second loop behind a run-time branch): measured, not kept.
                    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) double*)hull_lds);
                    if (A.kind == K_HULL) {
                        const long off = hull_hv(A) - m.hull_blob;
                        A.rad = (double)(lds0 + 8u * (unsigned)off);
                        A.h[0] = __builtin_bit_cast(double, reinterpret_cast<unsigned long long>(hull_lds + off));
                    }
                    if (Bc.kind == K_HULL) {
                        const long off = hull_hv(Bc) - m.hull_blob;
                        Bc.rad = (double)(lds0 + 8u * (unsigned)off);
                        Bc.h[0] = __builtin_bit_cast(double, reinterpret_cast<unsigned long long>(hull_lds + off));
                    }
                }
                if (prof) { double acc = 0.0; for (int e = 0; e < 3; ++e) acc += A.c[e] + Bc.c[e] + A.ax[0][e] + A.ax[2][e] + Bc.ax[0][e] + Bc.ax[2][e] + A.h[e] + Bc.h[e]; if (acc == 12345.678) mark_hit(b, mask_bits, mask_bytes); }
                NBK_STAMP(4);
                const double* cst = m.vp_cst + 4 * p;
                int verdict;
                if (NBK_DBG(m) & 64) { double acc = 0.0; for (int e = 0; e < 3; ++e) acc += A.c[e] + Bc.c[e] + A.ax[0][e] + A.ax[1][e] + A.ax[2][e] + Bc.ax[0][e] + Bc.ax[1][e] + Bc.ax[2][e] + A.h[e] + Bc.h[e]; verdict = (acc == 12345.678) ? 1 : 0; }
                else if (Bc.kind == K_PLANE) verdict = plane_collides(A, Bc, thr, cst[2]) ? 1 : 0;
                else {
                    // step 2 of the predicate in float64: the float32 broadphase only culled what is certainly free
                    tc = (thr + cst[0]) + cst[1];
                    const double rs = (tc + cst[2]) + cst[3];
                    double dc[3];
                    sub3(A.c, Bc.c, dc);
                    if (!(rs > 0.0) || !(dot3(dc, dc) < rs * rs)) verdict = 0;
                    else verdict = cores_collide_pre(A, Bc, tc);
                }
                if (verdict == 1) mark_hit(b, mask_bits, mask_bytes);
                pooled = verdict < 0;
            }
            NBK_STAMP(5);
            // ---- phase 2: the undecided lanes walk GJK on their own item, one iteration per trip ---------------------------
            // (a chunk has at most 64 undecided items and 64 lanes: nothing to redistribute, so no pool)
            bool have = pooled && !(NBK_DBG(m) & 8);
            if constexpr (MODE == 1) {
                GjkBool gb;
                gjkb_init(gb, A, Bc);
                while (__builtin_amdgcn_ballot_w64(have) != 0ull) {
                    if (have) {
                        const int r = gjkb_step(gb, A, Bc);
                        if (r != 0) { if (r == 2) mark_hit(b, mask_bits, mask_bytes); have = false; }
                    }
                }
            } else if constexpr (MODE == 3) {
                // tc > 0 everywhere: the boolean walk on the inflated core, then the distance iteration for what it leaves
                // undecided.  The kernel's time is set by its slowest items: with a cap of 20 steps some 10-2000 items per 1e6
                // configurations fell through to the distance iteration (~9 steps of 4.4 us each: 87 us, no better than running it
                // on everything); with 64 none does on the benchmark scene and the tail is the walk's own 38 steps (66 us).
                GjkBool gb;
                gjkb_init(gb, A, Bc);
                // hull cores skip the walk: every step scans a vertex list and the distance iteration needs half as many
                bool hard = have && (A.kind == K_HULL || Bc.kind == K_HULL);
                have = have && !hard;
                while (__builtin_amdgcn_ballot_w64(have) != 0ull) {
                    if (have) {
                        const int r = gjkb_step<2>(gb, A, Bc, tc);
                        if (r != 0) { if (r == 2) mark_hit(b, mask_bits, mask_bytes); hard = hard || r == 3; have = false; }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(hard) != 0ull) {
                    GjkPred g;
                    gjk_pred_init(g, A, Bc);
                    while (__builtin_amdgcn_ballot_w64(hard) != 0ull) {
                        if (hard) {
                            const int r = gjk_pred_step<true>(g, A, Bc, tc);      // (<true>: tc >= 0 here, no overlap-depth estimate)
                            if (r != 0) { if (r == 2) mark_hit(b, mask_bits, mask_bytes); hard = false; }
                        }
                    }
                }
            } else if constexpr (MODE == 2) {
                GjkPred g;
                gjk_pred_init(g, A, Bc);
                while (__builtin_amdgcn_ballot_w64(have) != 0ull) {
                    if (have) {
                        const int r = gjk_pred_step(g, A, Bc, tc);
                        if (r != 0) { if (r == 2) mark_hit(b, mask_bits, mask_bytes); have = false; }
                    }
                }
            } else {
                GjkBool gb;
                GjkPred g;
                gjkb_init(gb, A, Bc);
                gjk_pred_init(g, A, Bc);
                bool walk = tc >= 0.0 && A.kind != K_HULL && Bc.kind != K_HULL;    // (a NaN threshold takes the distance iteration)
                while (__builtin_amdgcn_ballot_w64(have) != 0ull) {
                    if (have) {
                        const int r = walk ? gjkb_step<2>(gb, A, Bc, tc) : gjk_pred_step(g, A, Bc, tc);
                        if (r == 3) walk = false;        // the inflated walk gave up: the distance iteration starts from its own (untouched) state
                        else if (r != 0) { if (r == 2) mark_hit(b, mask_bits, mask_bytes); have = false; }
                    }
                }
            }
        }
        __syncthreads();
        chunk += (unsigned)__builtin_amdgcn_readfirstlane((int)next_ticket);
        if (prof && i0 == (unsigned long long)part * NARROW_T) {
            NBK_STAMP(6);
            if (threadIdx.x == 0) {
                for (int e = 0; e < 6; ++e) atomicAdd(&g_narrow_prof[e], stamp[e + 1] - stamp[e]);
                atomicAdd(&g_narrow_prof[15], 1ull);
                atomicMax(&g_narrow_prof[14], stamp[6] - stamp[0]);
                atomicMin(&g_narrow_prof[13], stamp[0]);
                atomicMax(&g_narrow_prof[12], stamp[6]);
            }
        }
    }
}

__global__ __launch_bounds__(NARROW_T, NARROW_WAVES_BOOL) void k_narrow_bool(DevModel m, EdgeSrc es, const double* __restrict__ q, double thr,
                                                      const unsigned long long* __restrict__ q_items,
                                                      unsigned long long* q_count, unsigned long long cap,
                                                      uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
    unsigned long long* __restrict__ count_next) {
    extern __shared__ double qstage[];          // NARROW_T * n_q doubles
    narrow_body<1>(m, es, q, thr, q_items, q_count, cap, mask_bits, mask_bytes, qstage, count_next);
}

__global__ __launch_bounds__(NARROW_T, NARROW_WAVES_GEN) void k_narrow(DevModel m, EdgeSrc es, const double* __restrict__ q, double thr,
                                                 const unsigned long long* __restrict__ q_items,
                                                 unsigned long long* q_count, unsigned long long cap,
                                                 uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
    unsigned long long* __restrict__ count_next) {
    extern __shared__ double qstage[];          // NARROW_T * n_q doubles
    narrow_body<0>(m, es, q, thr, q_items, q_count, cap, mask_bits, mask_bytes, qstage, count_next);
}

__global__ __launch_bounds__(NARROW_T, NARROW_WAVES_GEN) void k_narrow_pred(DevModel m, EdgeSrc es, const double* __restrict__ q, double thr,
                                                      const unsigned long long* __restrict__ q_items,
                                                      unsigned long long* q_count, unsigned long long cap,
                                                      uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
    unsigned long long* __restrict__ count_next) {
    extern __shared__ double qstage[];          // NARROW_T * n_q doubles
    narrow_body<2>(m, es, q, thr, q_items, q_count, cap, mask_bits, mask_bytes, qstage, count_next);
}

__global__ __launch_bounds__(NARROW_T, 2) void k_narrow_pos(DevModel m, EdgeSrc es, const double* __restrict__ q, double thr,
                                                     const unsigned long long* __restrict__ q_items,
                                                     unsigned long long* q_count, unsigned long long cap,
                                                     uint64_t* __restrict__ mask_bits, uint8_t* __restrict__ mask_bytes,
    unsigned long long* __restrict__ count_next) {
    extern __shared__ double qstage[];          // NARROW_T * n_q doubles
    narrow_body<3>(m, es, q, thr, q_items, q_count, cap, mask_bits, mask_bytes, qstage, count_next);
}

// MODE 0: min distance + argmin; MODE 1: all pair distances; MODE 2: all pair distances + witnesses;
// MODE 3: distances + witnesses + proximity-Jacobian rows (Arm.jacobian_proximity, arm.py:620-632):
//         row[col(k)] = n . Jv_subject,k(p_s) - n . Jv_target,k(p_t) over the joints k above each shape, with
//         Jv_k(r) = w_k x (r - o_k) (revolute) or w_k (prismatic); world targets contribute nothing.
// one row of the proximity Jacobian (MODE 3) of pair p for the configuration parked in LDS column `col`
NBK_DEV void prox_row(const DevModel& m, const double* lds_jz, int col, int p, const double* wit, double* row) {
    for (int c = 0; c < m.n_q; ++c) row[c] = 0.0;
    const int sa = m.pair_a[p], sb = m.pair_b[p];
    const unsigned ma = m.rs_mask[sa];
    const unsigned mb = sb < m.n_rshapes ? m.rs_mask[sb] : 0u;
    for (int k = 0; k < m.n_joints; ++k) {
        const bool in_a = (ma >> k) & 1u, in_b = (mb >> k) & 1u;
        if (!in_a && !in_b) continue;
        const double w[3] = {lds_jz[(6 * k) * WAVE + col], lds_jz[(6 * k + 1) * WAVE + col], lds_jz[(6 * k + 2) * WAVE + col]};
        const double o3[3] = {lds_jz[(6 * k + 3) * WAVE + col], lds_jz[(6 * k + 4) * WAVE + col], lds_jz[(6 * k + 5) * WAVE + col]};
        const bool rev = m.joint_type[k] == NBK_REVOLUTE;
        double va = 0.0, vb = 0.0;
        if (in_a) {
            if (rev) { double dd[3], v[3]; sub3(wit, o3, dd); cross3(w, dd, v); va = dot3(wit + 6, v); }
            else va = dot3(wit + 6, w);
        }
        if (in_b) {
            if (rev) { double dd[3], v[3]; sub3(wit + 3, o3, dd); cross3(w, dd, v); vb = dot3(wit + 6, v); }
            else vb = dot3(wit + 6, w);
        }
        row[m.joint_qidx[k]] = va - vb;
    }
}

constexpr int EPAQ_CAP = 128;                  // (lane, pair) items waiting for their EPA pass, per wave

// MODE >= 1: TWO waves per workgroup share the parked cores of the block's 64 configurations (lane = configuration in both) and take
// every other pair of the workgroup's slice: a robot whose parked rows fill 70+ KB fits two workgroups per CU, and the kernel's
// 256 registers allow four waves -- with one-wave workgroups half the SIMDs idled (mesh scene: records 7.3 -> see DESIGN.md).  Both
// waves stage q and sweep the tree themselves (identical values to identical LDS addresses; no hand-over to wait for).
constexpr int EPAQ_DOUBLES = EPAQ_CAP + EPAQ_CAP / 2;        // one wave's EPA queue: depths [EPAQ_CAP] double, items [EPAQ_CAP] unsigned
template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 64 : 128) void k_distances(DevModel m, const double* __restrict__ q, int64_t B,
                                                   double* __restrict__ out_d, int32_t* __restrict__ out_i,
                                                   double* __restrict__ out_w, double* __restrict__ out_j = nullptr) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x % WAVE;
    const int wave = MODE == 0 ? 0 : (int)(threadIdx.x / WAVE);
    constexpr int NWAVE = MODE == 0 ? 1 : 2;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    double* lds_q = lds;
    double* lds_s = lds_q + WAVE * m.n_q;
    double* lds_fr = lds_s + WAVE * m.shape_rows;
    stage_q(q, base, B, m.n_q, lds_s, lds_q, lane);
    const int64_t b = base + lane;
    const bool active = b < B;
    double* lds_jz = (MODE == 3) ? lds_fr + WAVE * 12 * m.frame_slots : nullptr;
    sweep_and_park(m, lds_q, lds_s, lds_fr, lane, lds_jz);
    double best = NBK_INF;
    int bi = -1;
    // MODE >= 1: overlapping cores whose exact depth needs EPA (a cylinder or a hull) get the axis-family value first and are queued
    // (lane, pair, family depth) in the LDS tail the validity path uses for its own queue; the queue is drained one item per lane --
    // 64 polytopes grow side by side instead of one lane's while 63 wait -- and a better (smaller) depth overwrites the record.
    double* epaq_depth = lds_fr + WAVE * 12 * m.frame_slots + (MODE == 3 ? WAVE * 6 * m.n_joints : 0) + wave * EPAQ_DOUBLES;
    unsigned* epaq_item = reinterpret_cast<unsigned*>(epaq_depth + EPAQ_CAP);
    int epaq_n = 0;
    auto epaq_drain = [&]() {
#ifdef NBK_NO_EPA_DRAIN
        epaq_n = 0; return;
#endif
        wave_lds_sync();
        for (int i0 = 0; i0 < epaq_n; i0 += WAVE) {
            const int i = i0 + lane;
            if (i < epaq_n) {
                const unsigned item = epaq_item[i];
                const int src = (int)(item & 63u), p = (int)(item >> 6);
                const double fam = epaq_depth[i];
                Core A, Bc;
                const int a = m.pair_a[p], bb = m.pair_b[p];
                load_core_any(m, lds_s, a, src, A);
                load_core_any(m, lds_s, bb < m.n_rshapes ? bb : ~(bb - m.n_rshapes), src, Bc);
                double o[4];
                if (epa_depth_copy(A, Bc, o, 0, 0.0) == 1 && o[0] < fam) {
                    // the record of cores_distance's overlap branch, with EPA's depth and direction
                    const double n[3] = {o[1], o[2], o[3]};
                    const double dc = -o[0];
                    const int64_t bs = base + src;
                    const int64_t oo = bs * m.n_pairs + m.pair_user[p];
                    out_d[oo] = (dc - A.margin) - Bc.margin;
                    if constexpr (MODE >= 2) {
                        const double neg[3] = {-n[0], -n[1], -n[2]};
                        double pa[3], pb[3], wit[9];
                        core_support(A, neg, pa);
                        axpy3(dc, n, pa, pb);
                        axpy3(-A.margin, n, pa, wit);
                        axpy3(Bc.margin, n, pb, wit + 3);
                        copy3(n, wit + 6);
#pragma unroll
                        for (int e = 0; e < 9; ++e) out_w[oo * 9 + e] = wit[e];
                        if constexpr (MODE == 3) prox_row(m, lds_jz, src, p, wit, out_j + oo * m.n_q);
                    }
                }
            }
        }
        epaq_n = 0;
        wave_lds_sync();
    };
    // MODE >= 1: gridDim.y workgroups share the pairs of a block of configurations (each sweeps the tree itself and takes every
    // gridDim.y-th ... contiguous slice of the pair list): small batches such as IRIS' 10 071 samples would otherwise put one
    // wave per 64 samples on a 1 024-SIMD chip and walk 44 GJK distances one after the other
    const int p_lo = (MODE == 0) ? 0 : (int)(((long long)m.n_pairs * blockIdx.y) / gridDim.y);
    const int p_hi = (MODE == 0) ? m.n_pairs : (int)(((long long)m.n_pairs * (blockIdx.y + 1)) / gridDim.y);
    for (int p = p_lo + wave; p < p_hi; p += NWAVE) {
        Core A, Bc;
        load_pair(m, lds_s, p, lane, A, Bc);
        // (every lane holds the same two shapes here: a hull's vertices go through the scalar cache, see core_support)
        if (A.kind == K_HULL) A.rad = -1.0;
        if (Bc.kind == K_HULL) Bc.rad = -1.0;
        double wit[9];
        double fam = -1.0;
        const double d = cores_distance<(MODE >= 2), (MODE >= 1)>(A, Bc, wit, &fam);
        if constexpr (MODE == 0) {
            // pairs are visited in device order; ties resolve to the smallest USER index like the oracle
            const int u = m.pair_user[p];
            if (d < best || (d == best && u < bi)) { best = d; bi = u; }
        } else {
            if (active) {
                const int64_t o = b * m.n_pairs + m.pair_user[p];
                out_d[o] = d;
                if constexpr (MODE >= 2) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) out_w[o * 9 + e] = wit[e];
                }
                if constexpr (MODE == 3) prox_row(m, lds_jz, lane, p, wit, out_j + o * m.n_q);
            }
            const bool need = active && fam >= 0.0;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(need);
            if (bal != 0ull) {
                if (need) {
                    const int pos = epaq_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    epaq_item[pos] = ((unsigned)p << 6) | (unsigned)lane;
                    epaq_depth[pos] = fam;
                }
                epaq_n += __builtin_popcountll(bal);
                if (epaq_n > EPAQ_CAP - WAVE) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the provisional records are out before a drain overwrites them
                    epaq_drain();
                }
            }
        }
    }
    if constexpr (MODE == 0) {
        if (active) { out_d[b] = best; if (out_i != nullptr) out_i[b] = bi; }
    } else {
        if (epaq_n > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            epaq_drain();
        }
    }
}

// ---- closest pair with branch-and-bound ----------------------------------------------------------------------
// min over the allowed pairs of the exact signed distance, and the first pair attaining it.  Cheap bounds first:
// with D = |cA - cB| (centres are points of the cores), D - (mA + mB) >= d_p >= D - (rhoA + rhoB) - (mA + mB),
// so only pairs whose lower bound does not exceed the smallest upper bound U can be the minimum.  Those few
// (lane, pair) items are compacted into an LDS queue and their exact distances (GJK converged to 1e-10) are
// evaluated on full waves; every lane then reduces its own items (minimum, ties to the smallest pair index).
// The result is exactly that of evaluating every pair (k_distances<0>): only work is skipped.
constexpr int CQ_CAP = 512;

NBK_DEV unsigned long long orderable(double d) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
NBK_DEV double from_orderable(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __builtin_bit_cast(double, u);
}

// D^2 (or the plane height hc) and the two constants R = rho sum + margin sum, M = margin sum of pair p for this lane
NBK_DEV void closest_bounds(const DevModel& m, const double* lds_s, int p, int lane, double& dd, double& hc, bool& plane, double& R, double& M) {
    const int a = m.pair_a[p], b = m.pair_b[p];
    const double* ra = lds_s + m.rs_row[a] * WAVE + lane;
    const double ca[3] = {ra[0], ra[WAVE], ra[2 * WAVE]};
    const double mA = m.rs_core[6 * a + 4], rhoA = m.rs_core[6 * a + 5];
    plane = false; hc = 0.0; dd = 0.0;
    if (b < m.n_rshapes) {
        const double* rb = lds_s + m.rs_row[b] * WAVE + lane;
        const double d[3] = {ca[0] - rb[0], ca[1] - rb[WAVE], ca[2] - rb[2 * WAVE]};
        dd = dot3(d, d);
        M = mA + m.rs_core[6 * b + 4];
        R = (rhoA + m.rs_core[6 * b + 5]) + M;
    } else {
        const int w = b - m.n_rshapes;
        const double* wc = m.ws_core + 18 * w;
        const double d[3] = {ca[0] - wc[0], ca[1] - wc[1], ca[2] - wc[2]};
        if (m.ws_kind[w] == K_PLANE) {
            const double n[3] = {wc[9], wc[10], wc[11]};
            plane = true;
            hc = dot3(d, n);
            M = mA;
            R = rhoA + mA;
        } else {
            dd = dot3(d, d);
            M = mA + wc[16];
            R = (rhoA + wc[17]) + M;
        }
    }
}

__global__ __launch_bounds__(64) void k_closest(DevModel m, const double* __restrict__ q, int64_t B,
                                                 double* __restrict__ out_d, int32_t* __restrict__ out_i) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    double* lds_q = lds;
    double* lds_s = lds_q + WAVE * m.n_q;
    double* lds_fr = lds_s + WAVE * m.shape_rows;
    double* lds_res = lds_fr + WAVE * 12 * m.frame_slots;                               // [CQ_CAP]
    unsigned long long* lds_best = reinterpret_cast<unsigned long long*>(lds_res + CQ_CAP);   // [64]
    unsigned* lds_arg = reinterpret_cast<unsigned*>(lds_best + WAVE);                   // [64]
    unsigned* queue = lds_arg + WAVE;                                                    // [CQ_CAP]
    unsigned short* elist = reinterpret_cast<unsigned short*>(queue + CQ_CAP);                                                    // [CQ_CAP] queue slots (of pass 2) whose depth needs EPA
    constexpr unsigned CQ_DEFER = 0x40000000u;                                           // queue item flag: provisional (axis-family) depth
    stage_q(q, base, B, m.n_q, lds_s, lds_q, lane);
    const int64_t b = base + lane;
    const bool active = b < B;
    sweep_and_park(m, lds_q, lds_s, lds_fr, lane);
    const int P = m.n_pairs;
    if (P == 0) {
        if (active) { out_d[b] = NBK_INF; if (out_i != nullptr) out_i[b] = -1; }
        return;
    }
    // pass 1: smallest upper bound U0 and the pair with the smallest lower bound (float square roots, rounded
    // outwards, are enough for bounds)
    double U = NBK_INF, lb1 = NBK_INF;
    int p1 = 0;
    for (int p = 0; p < P; ++p) {
        double dd, hc, R, M; bool plane;
        closest_bounds(m, lds_s, p, lane, dd, hc, plane, R, M);
        const float sf = __builtin_sqrtf((float)dd);
        const double Dhi = plane ? hc : (double)(sf * 1.000001f + 1e-30f);
        const double Dlo = plane ? hc : (double)(sf * 0.999999f);
        const double ub = Dhi - M, lb = Dlo - R;
        U = ub < U ? ub : U;
        if (lb < lb1) { lb1 = lb; p1 = p; }
    }
    lds_best[lane] = ~0ull;
    lds_arg[lane] = 0x7FFFFFFFu;
    // round 1 (dense, one item per lane): the exact distance of that most promising pair -- EPA inline where the cores overlap: its
    // depth is what makes U tight for a colliding configuration -- tightens U a lot
    queue[lane] = active ? (((unsigned)p1 << 6) | (unsigned)lane) : 0xFFFFFFFFu;
    {
        double d1 = NBK_INF;
        if (active) {
            const int a = m.pair_a[p1], bb = m.pair_b[p1];
            Core A, Bc;
            load_core_any(m, lds_s, a, lane, A);
            load_core_any(m, lds_s, bb < m.n_rshapes ? bb : ~(bb - m.n_rshapes), lane, Bc);
            d1 = cores_distance<false>(A, Bc, nullptr);
            lds_best[lane] = orderable(d1);
        }
        lds_res[lane] = d1;
        U = d1 < U ? d1 : U;
    }
    const double Ucut = U + (1e-9 + 1e-9 * __builtin_fabs(U));
    // pass 2: the remaining candidates -> queue (slots 64..)
    int qn = WAVE;
    bool overflow = false;
    for (int p = 0; p < P; ++p) {
        double dd, hc, R, M; bool plane;
        closest_bounds(m, lds_s, p, lane, dd, hc, plane, R, M);
        bool cand;
        if (plane) cand = (hc - R) <= Ucut;
        else { const double t = Ucut + R; cand = (t >= 0.0) && (dd <= t * t); }
        cand = cand && active && (p != p1);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(cand);
        if (bal == 0ull) continue;
        const int cnt = __builtin_popcountll(bal);
        if (qn + cnt > CQ_CAP) { overflow = true; break; }
        if (cand) {
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            queue[pos] = ((unsigned)p << 6) | (unsigned)lane;
        }
        qn += cnt;
    }
    if (overflow) {
        // (never seen on the test scenes) every pair, as k_distances<0> does
        double best = NBK_INF;
        int bi = -1;
        for (int p = 0; p < P; ++p) {
            Core A, Bc;
            load_pair(m, lds_s, p, lane, A, Bc);
            const double d = cores_distance<false>(A, Bc, nullptr);
            const int u = m.pair_user[p];
            if (d < best || (d == best && u < bi)) { best = d; bi = u; }
        }
        if (active) { out_d[b] = best; if (out_i != nullptr) out_i[b] = bi; }
        return;
    }
    __syncthreads();
    for (int i0 = WAVE; i0 < qn; i0 += WAVE) {
        const int i = i0 + lane;
        if (i < qn) {
            const unsigned item = queue[i];
            const int src = (int)(item & 63u), p = (int)(item >> 6);
            const int a = m.pair_a[p], bb = m.pair_b[p];
            Core A, Bc;
            load_core_any(m, lds_s, a, src, A);
            load_core_any(m, lds_s, bb < m.n_rshapes ? bb : ~(bb - m.n_rshapes), src, Bc);
            double fam = -1.0;
            const double d = cores_distance<false, true>(A, Bc, nullptr, &fam);
            lds_res[i] = d;
            if (fam >= 0.0) queue[i] = item | CQ_DEFER;
            else atomicMin(&lds_best[src], orderable(d));
        }
    }
    __syncthreads();
    // the deferred items, compacted; one whose provisional value (a lower bound) is already above its lane's best cannot be the
    // minimum nor tie with it and is dropped -- the best only decreases, so a stale read errs on the side of running EPA
    int en = 0;
    for (int i0 = 0; i0 < qn; i0 += WAVE) {
        const int i = i0 + lane;
        bool need = false;
        if (i < qn) {
            const unsigned item = queue[i];
            need = item != 0xFFFFFFFFu && (item & CQ_DEFER) != 0u && !(orderable(lds_res[i]) > lds_best[(int)(item & 63u)]);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(need);
        if (need) elist[en + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (unsigned short)i;
        en += __builtin_popcountll(bal);
    }
    __syncthreads();
    for (int j0 = 0; j0 < en; j0 += WAVE) {
        const int j = j0 + lane;
        if (j < en) {
            const int i = (int)elist[j];
            const unsigned item = queue[i] & ~CQ_DEFER;
            const int src = (int)(item & 63u), p = (int)(item >> 6);
            const int a = m.pair_a[p], bb = m.pair_b[p];
            Core A, Bc;
            load_core_any(m, lds_s, a, src, A);
            load_core_any(m, lds_s, bb < m.n_rshapes ? bb : ~(bb - m.n_rshapes), src, Bc);
            // overlap_depth_exact, in two steps: the family value again (4 KB of LDS to keep it cost a resident workgroup on the
            // benchmark arm), then EPA's if it is smaller
            double n[3], o[4];
            double depth = overlap_depth(A, Bc, n);
            if (epa_depth_copy(A, Bc, o, 0, 0.0) == 1 && o[0] < depth) depth = o[0];
            const double d = ((-depth) - A.margin) - Bc.margin;
            lds_res[i] = d;
            atomicMin(&lds_best[src], orderable(d));
        }
    }
    __syncthreads();
    for (int i0 = 0; i0 < qn; i0 += WAVE) {
        const int i = i0 + lane;
        if (i < qn) {
            const unsigned item = queue[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (queue[i] & ~CQ_DEFER);
            if (item != 0xFFFFFFFFu) {
                const int src = (int)(item & 63u), p = (int)(item >> 6);
                if (orderable(lds_res[i]) == lds_best[src]) atomicMin(&lds_arg[src], (unsigned)m.pair_user[p]);
            }
        }
    }
    __syncthreads();
    if (active) {
        const unsigned long long k = lds_best[lane];
        out_d[b] = (k == ~0ull) ? NBK_INF : from_orderable(k);
        if (out_i != nullptr) out_i[b] = (k == ~0ull) ? -1 : (int32_t)lds_arg[lane];
    }
}

// ---- DiscreteConnector: one edge per wave, lanes = interpolation samples ---------------------------
__global__ __launch_bounds__(64) void k_edges(DevModel m, const double* __restrict__ starts, const double* __restrict__ goals,
                                               const double* __restrict__ dist, int64_t E, double resolution,
                                               double max_distance, int mode, double thr, uint8_t* __restrict__ valid,
                                               double* __restrict__ end, int32_t* __restrict__ n_samples,
                                               const uint8_t* __restrict__ only, unsigned long long* __restrict__ stats) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int64_t e = blockIdx.x;
    // `only`: serve just the edges the flat batch could not hold (k_edge_expand's overflow marks)
    if (only != nullptr) {
        if (!only[e]) return;
        if (lane == 0 && stats != nullptr) atomicAdd(stats + 1, 1ull);
    }
    const int nq = m.n_q;
    double* lds_q = lds;
    double* lds_s = lds_q + WAVE * nq;
    double* lds_fr = lds_s + WAVE * m.shape_rows;
    unsigned* lds_x = reinterpret_cast<unsigned*>(lds_fr + WAVE * 12 * m.frame_slots);
    const double* s = starts + e * nq;
    const double* g = goals + e * nq;
    double d;
    if (dist != nullptr) d = dist[e];
    else {
        double acc = 0.0;
        for (int i = 0; i < nq; ++i) { const double df = g[i] - s[i]; acc = NBK_FMA(df, df, acc); }
        d = nbk_sqrt(acc);
    }
    if (!(d > 1.1920928955078125e-07 && d <= 1.7976931348623157e308)) {   // float32 eps: the reference returns None
        if (lane == 0) { valid[e] = 0; if (n_samples) n_samples[e] = 0; }
        if (end != nullptr && lane < nq) end[e * nq + lane] = __builtin_nan("");
        return;
    }
    const double Tf = (mode == NBK_STEER && d > max_distance) ? max_distance / d : 1.0;
    const double step = resolution / d;
    const double lenf = __builtin_ceil(Tf / step);
    const int64_t n = lenf > 0.0 ? (int64_t)lenf : 0;     // len(arange(0, Tf, step)); samples = n + 1
    bool ok = true;
    for (int64_t c0 = 0; c0 <= n && ok; c0 += WAVE) {
        const int64_t i = c0 + lane;
        const bool active = i <= n;
        const double t = i < n ? (double)i * step : Tf;
        const double omt = 1.0 - t;
        for (int j = 0; j < nq; ++j) {
            const double a = omt * s[j];
            const double bb = t * g[j];
            lds_q[j * WAVE + lane] = a + bb;
        }
        const bool bad = row_nonfinite(lds_q + lane, nq, WAVE);
        sweep_and_park(m, lds_q, lds_s, lds_fr, lane);
        const bool hit = wave_collides(m, lds_s, lds_x, lane, thr, active) || bad;
        if (__builtin_amdgcn_ballot_w64(hit && active) != 0ull) ok = false;
    }
    if (lane == 0) { valid[e] = ok ? 1 : 0; if (n_samples) n_samples[e] = (int32_t)(n + 1); }
    if (end != nullptr && lane < nq) {
        if (mode == NBK_CONNECT) end[e * nq + lane] = g[lane];
        else {
            const double omt = 1.0 - Tf;
            const double a = omt * s[lane];
            const double bb = Tf * g[lane];
            end[e * nq + lane] = a + bb;
        }
    }
}

__global__ void k_selftest(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* so, double* co,
                           double* sq, double* dv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    nbk_sincos(a[i], s, c);
    so[i] = s; co[i] = c;
    sq[i] = nbk_sqrt(a[i]);
    dv[i] = a[i] / b[i];
}

// ==== batched DiscreteConnector over many edges: plan -> scan -> expand -> validity pipeline -> reduce ======
// Every sample of every edge becomes one configuration of a flat batch that goes through k_broad / k_narrow
// (q is generated on the fly from the edge's end points), so edges run at the batch-validity rate; the
// one-wave-per-edge kernel above stays for a handful of edges, where its early exit and single launch win.
__global__ void k_edge_plan(int nq, const double* __restrict__ starts, const double* __restrict__ goals,
                            const double* __restrict__ dist, int64_t E, double resolution, double max_distance, int mode,
                            double* __restrict__ plan, unsigned long long* __restrict__ cnt, double* __restrict__ end,
                            int32_t* __restrict__ n_samples) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const double* s = starts + e * nq;
    const double* g = goals + e * nq;
    double d;
    if (dist != nullptr) d = dist[e];
    else {
        double acc = 0.0;
        for (int i = 0; i < nq; ++i) { const double df = g[i] - s[i]; acc = NBK_FMA(df, df, acc); }
        d = nbk_sqrt(acc);
    }
    if (!(d > 1.1920928955078125e-07 && d <= 1.7976931348623157e308)) {
        plan[3 * e] = 0.0; plan[3 * e + 1] = 0.0; plan[3 * e + 2] = 0.0;
        cnt[e] = 0ull;
        if (n_samples) n_samples[e] = 0;
        if (end) for (int i = 0; i < nq; ++i) end[e * nq + i] = __builtin_nan("");
        return;
    }
    const double Tf = (mode == NBK_STEER && d > max_distance) ? max_distance / d : 1.0;
    const double step = resolution / d;
    const double lenf = __builtin_ceil(Tf / step);
    const long long n = lenf > 0.0 ? (long long)lenf : 0;
    plan[3 * e] = step; plan[3 * e + 1] = Tf; plan[3 * e + 2] = (double)n;
    cnt[e] = (unsigned long long)(n + 1);
    if (n_samples) n_samples[e] = (int32_t)(n + 1);
    if (end) {
        if (mode == NBK_CONNECT) for (int i = 0; i < nq; ++i) end[e * nq + i] = g[i];
        else {
            const double omt = 1.0 - Tf;
            for (int i = 0; i < nq; ++i) { const double a = omt * s[i]; const double bb = Tf * g[i]; end[e * nq + i] = a + bb; }
        }
    }
}

// exclusive prefix sum of cnt[0..E) into offs[0..E], one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void k_scan(const unsigned long long* __restrict__ cnt, int64_t E, unsigned long long* __restrict__ offs) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int64_t chunk = (E + 1023) / 1024;
    const int64_t lo = (int64_t)t * chunk, hi = (lo + chunk) < E ? (lo + chunk) : E;
    unsigned long long acc = 0;
    for (int64_t i = lo; i < hi; ++i) acc += cnt[i];
    part[t] = acc;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        offs[E] = run;
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (int64_t i = lo; i < hi; ++i) { offs[i] = run; run += cnt[i]; }
}

// sample map of the flat batch; an edge that does not fit the scratch's capacity completely is marked (ovf) and left to the
// one-wave-per-edge kernel
__global__ __launch_bounds__(64) void k_edge_expand(const unsigned long long* __restrict__ offs, int64_t E, unsigned long long* __restrict__ map,
                                                     unsigned long long cap, uint8_t* __restrict__ ovf) {
    const int64_t e = blockIdx.x;
    const unsigned long long o = offs[e], hi = offs[e + 1];
    const bool over = hi > cap;
    if (threadIdx.x == 0) ovf[e] = over ? 1 : 0;
    const unsigned long long n = (hi < cap ? hi : cap) - (o < cap ? o : cap);
    for (unsigned long long i = threadIdx.x; i < n; i += WAVE) map[o + i] = ((unsigned long long)e << 32) | i;
}

__global__ __launch_bounds__(64) void k_edge_reduce(const unsigned long long* __restrict__ offs, int64_t E, const uint64_t* __restrict__ words,
                                                     const uint8_t* __restrict__ ovf, uint8_t* __restrict__ valid,
                                                     unsigned long long* __restrict__ stats) {
    const int64_t e = blockIdx.x;
    // the sample count this call needed goes to pinned host memory: the next call sizes its scratch from it
    if (e == 0 && threadIdx.x == 0 && stats != nullptr) { stats[0] = offs[E]; stats[1] = 0ull; }
    if (ovf[e]) return;                      // walked by k_edges
    const unsigned long long o = offs[e], n = offs[e + 1] - o;
    bool hit = false;
    for (unsigned long long i = threadIdx.x; i < n; i += WAVE) {
        const unsigned long long b = o + i;
        hit = hit || ((words[b >> 6] >> (b & 63)) & 1ull);
    }
    const unsigned long long any = __builtin_amdgcn_ballot_w64(hit);
    if (threadIdx.x == 0) valid[e] = (n > 0 && any == 0ull) ? 1 : 0;
}

// ---- host side --------------------------------------------------------------------------------------
struct Blob {
    std::vector<unsigned char> bytes;
    size_t add(const void* p, size_t n) {
        size_t off = (bytes.size() + 15) & ~size_t(15);
        bytes.resize(off + n);
        if (n) memcpy(bytes.data() + off, p, n);
        return off;
    }
};

static void core_params(int type, const double* param, int& kind, double* cc) {
    cc[0] = cc[1] = cc[2] = cc[3] = cc[4] = 0.0;
    switch (type) {
        case NBK_SPHERE: kind = K_POINT; cc[4] = param[0]; break;
        case NBK_CAPSULE: kind = K_SEG; cc[4] = param[0]; cc[0] = param[1]; break;
        case NBK_BOX:
            kind = K_BOX; cc[4] = param[3];
            cc[0] = param[0] - param[3]; cc[1] = param[1] - param[3]; cc[2] = param[2] - param[3];
            break;
        case NBK_CYLINDER: kind = K_CYL; cc[4] = param[3]; cc[3] = param[0] - param[3]; cc[0] = param[1] - param[3]; break;
        case NBK_HULL: kind = K_HULL; cc[4] = param[3]; break;      // cc[0..2] (the HullRef) and the radius are filled in by the caller
        default: kind = K_PLANE; break;
    }
}

static double host_bound_radius(int kind, const double* cc) {
    // must round exactly like the oracle's core_bound_radius
    switch (kind) {
        case K_POINT: return 0.0;
        case K_SEG: return cc[0];
        case K_CYL: return sqrt(fma(cc[3], cc[3], cc[0] * cc[0]));
        case K_BOX: return sqrt(fma(cc[2], cc[2], fma(cc[1], cc[1], cc[0] * cc[0])));
        case K_HULL: return cc[5];                 // stored: hull_bound_radius of its vertices
        default: return HUGE_VAL;
    }
}

// largest vertex norm of a hull; must round exactly like the oracle's hull_bound_radius
static double hull_bound_radius(const double* v, int n) {
    double best = 0.0;
    for (int k = 0; k < n; ++k) {
        const double r2 = fma(v[3 * k + 2], v[3 * k + 2], fma(v[3 * k + 1], v[3 * k + 1], v[3 * k] * v[3 * k]));
        if (r2 > best) best = r2;
    }
    return sqrt(best);
}

static int host_core_rows(int kind) { return kind == K_POINT ? 3 : ((kind == K_BOX || kind == K_HULL) ? 12 : 6); }

}  // namespace nbk

using namespace nbk;

extern "C" {

int32_t nbk_abi_version(void) { return NBK_ABI_VERSION; }

const char* nbk_status_string(int32_t st) {
    switch (st) {
        case NBK_OK: return "ok";
        case NBK_ERR_INVALID: return "invalid argument";
        case NBK_ERR_NO_DEVICE: return "no HIP device available";
        case NBK_ERR_HIP: return "HIP runtime error";
        case NBK_ERR_UNSUPPORTED: return "descriptor exceeds a compiled-in limit";
        case NBK_ERR_ALLOC: return "allocation failed";
        default: return "unknown status";
    }
}

const char* nbk_last_error(void) { return g_err; }

int32_t nbk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int32_t nbk_model_create(const nbk_model_desc* d, nbk_model** out) {
    if (d == nullptr || out == nullptr) return NBK_ERR_INVALID;
    *out = nullptr;
    if (nbk_device_count() <= 0) return NBK_ERR_NO_DEVICE;
    const int J = d->n_joints, S = d->n_rshapes, W = d->n_wshapes, P = d->n_pairs;
    if (d->n_q < 0 || J < 0 || S < 0 || W < 0 || P < 0) return NBK_ERR_INVALID;
    if (J > NBK_MAX_JOINTS || d->n_q > NBK_MAX_DOF) return NBK_ERR_UNSUPPORTED;
    const int H = d->n_hulls;
    if (H < 0 || (H > 0 && (d->hull_vert_begin == nullptr || d->hull_verts == nullptr || d->hull_face_begin == nullptr))) return NBK_ERR_INVALID;
    for (int h = 0; h < H; ++h) {
        if (d->hull_vert_begin[h + 1] <= d->hull_vert_begin[h] || d->hull_face_begin[h + 1] < d->hull_face_begin[h]) return NBK_ERR_INVALID;
        if (h == 0 && (d->hull_vert_begin[0] != 0 || d->hull_face_begin[0] != 0)) return NBK_ERR_INVALID;
    }
    if (H > 0 && d->hull_face_begin[H] > 0 && d->hull_planes == nullptr) return NBK_ERR_INVALID;
    // face planes: unit outward normals that bound the vertex set (n.v <= d for every vertex).  The float32 broadphase certifies
    // hits from the ball the planes inscribe and the overlap depth walks them: planes that are not what nbk.h asks for would
    // produce verdicts neither the narrowphase nor the oracle ever re-examines.
    for (int h = 0; h < H; ++h) {
        const double* v = d->hull_verts + 3 * (size_t)d->hull_vert_begin[h];
        const int nv = d->hull_vert_begin[h + 1] - d->hull_vert_begin[h];
        double vmax = 0.0;
        for (int i = 0; i < 3 * nv; ++i) { if (!(fabs(v[i]) <= 1e300)) return NBK_ERR_INVALID; vmax = std::max(vmax, fabs(v[i])); }
        for (int f = d->hull_face_begin[h]; f < d->hull_face_begin[h + 1]; ++f) {
            const double* pl = d->hull_planes + 4 * (size_t)f;
            const double n2 = pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2];
            if (!(fabs(n2 - 1.0) <= 1e-9) || !(fabs(pl[3]) <= 1e300)) {
                snprintf(g_err, sizeof(g_err), "hull %d, plane %d: the normal must have unit length (|n|^2 = %.17g)", h, f - d->hull_face_begin[h], n2);
                return NBK_ERR_INVALID;
            }
            const double tol = 1e-9 * (1.0 + fabs(pl[3]) + vmax);
            for (int i = 0; i < nv; ++i)
                if (pl[0] * v[3 * i] + pl[1] * v[3 * i + 1] + pl[2] * v[3 * i + 2] > pl[3] + tol) {
                    snprintf(g_err, sizeof(g_err), "hull %d, plane %d does not bound vertex %d (n.v - d = %.3g)", h, f - d->hull_face_begin[h], i,
                             pl[0] * v[3 * i] + pl[1] * v[3 * i + 1] + pl[2] * v[3 * i + 2] - pl[3]);
                    return NBK_ERR_INVALID;
                }
        }
    }
    auto hull_ok = [&](double idx) { return idx >= 0.0 && idx < (double)H && idx == (double)(int)idx; };
    for (int k = 0; k < J; ++k) {
        if (d->joint_parent[k] >= k || d->joint_parent[k] < -1) return NBK_ERR_INVALID;   // parents first
        if (d->joint_qidx[k] < 0 || d->joint_qidx[k] >= d->n_q) return NBK_ERR_INVALID;
        if (d->joint_type[k] != NBK_REVOLUTE && d->joint_type[k] != NBK_PRISMATIC) return NBK_ERR_INVALID;
    }
    for (int s = 0; s < S; ++s) {
        if (d->rshape_frame[s] < -1 || d->rshape_frame[s] >= J) return NBK_ERR_INVALID;
        const int t = d->rshape_type[s];
        if (!((t >= NBK_SPHERE && t <= NBK_CYLINDER) || t == NBK_HULL)) return NBK_ERR_INVALID;
        if (t == NBK_HULL && !hull_ok(d->rshape_param[4 * s])) return NBK_ERR_INVALID;
    }
    for (int w = 0; w < W; ++w) {
        if (d->wshape_type[w] < NBK_SPHERE || d->wshape_type[w] > NBK_HULL) return NBK_ERR_INVALID;
        if (d->wshape_type[w] == NBK_HULL && !hull_ok(d->wshape_param[4 * w])) return NBK_ERR_INVALID;
    }
    for (int p = 0; p < P; ++p) {
        if (d->pair_a[p] < 0 || d->pair_a[p] >= S) return NBK_ERR_INVALID;
        if (d->pair_b[p] < 0 || d->pair_b[p] >= S + W) return NBK_ERR_INVALID;
    }
    // frame load/save plan: a frame stays in registers when its child is the next joint
    std::vector<int> load(J), save(J, -1);
    int slots = 0;
    for (int k = 0; k < J; ++k) {
        const int par = d->joint_parent[k];
        if (par == k - 1) load[k] = (par < 0) ? -1 : -2;
        else if (par < 0) load[k] = -1;
        else {
            if (save[par] < 0) save[par] = slots++;
            load[k] = save[par];
        }
    }
    // robot shapes in frame order
    std::vector<int> order;
    std::vector<int> begin(J + 2, 0);
    for (int f = -1; f < J; ++f) {
        begin[f + 1] = (int)order.size();
        for (int s = 0; s < S; ++s) if (d->rshape_frame[s] == f) order.push_back(s);
    }
    begin[J + 1] = (int)order.size();
    std::vector<int> new_index(S);
    std::vector<int> rs_kind(S), rs_row(S);
    std::vector<double> rs_local(12 * (size_t)S), rs_core(6 * (size_t)S);
    std::vector<int> rs_hull(S > 0 ? S : 1, -1), ws_hull(W > 0 ? W : 1, -1);      // hull index of K_HULL shapes
    int rows = 0;
    for (int i = 0; i < S; ++i) {
        const int s = order[i];
        new_index[s] = i;
        int kind;
        core_params(d->rshape_type[s], d->rshape_param + 4 * s, kind, &rs_core[6 * i]);
        if (kind == K_HULL) {
            const int h = (int)d->rshape_param[4 * s];
            rs_core[6 * i + 5] = hull_bound_radius(d->hull_verts + 3 * (size_t)d->hull_vert_begin[h], d->hull_vert_begin[h + 1] - d->hull_vert_begin[h]);
            rs_hull[i] = h;
        }
        rs_core[6 * i + 5] = host_bound_radius(kind, &rs_core[6 * i]);
        rs_kind[i] = kind;
        rs_row[i] = rows;
        rows += host_core_rows(kind);
        memcpy(&rs_local[12 * i], d->rshape_local + 12 * s, 12 * sizeof(double));
    }
    std::vector<int> ws_kind(W);
    bool world_hulls = false;
    std::vector<double> ws_core(18 * (size_t)W);
    for (int w = 0; w < W; ++w) {
        double cc[6];
        int kind;
        core_params(d->wshape_type[w], d->wshape_param + 4 * w, kind, cc);
        cc[5] = 0.0;
        if (kind == K_HULL) {
            const int h = (int)d->wshape_param[4 * w];
            cc[5] = hull_bound_radius(d->hull_verts + 3 * (size_t)d->hull_vert_begin[h], d->hull_vert_begin[h + 1] - d->hull_vert_begin[h]);
            ws_hull[w] = h;
        }
        ws_kind[w] = kind;
        world_hulls = world_hulls || kind == K_HULL;
        const double* T = d->wshape_pose + 12 * w;
        double* o = &ws_core[18 * w];
        o[0] = T[3]; o[1] = T[7]; o[2] = T[11];
        for (int j = 0; j < 3; ++j) { o[3 + 3 * j] = T[j]; o[4 + 3 * j] = T[4 + j]; o[5 + 3 * j] = T[8 + j]; }
        if (kind == K_PLANE) { o[9] = d->wshape_param[4 * w]; o[10] = d->wshape_param[4 * w + 1]; o[11] = d->wshape_param[4 * w + 2]; }
        o[12] = cc[0]; o[13] = cc[1]; o[14] = cc[2]; o[15] = cc[3]; o[16] = cc[4];
        o[17] = host_bound_radius(kind, cc);
    }
    std::vector<int> pa(P), pb(P), pu(P);
    for (int p = 0; p < P; ++p) {
        pa[p] = new_index[d->pair_a[p]];
        pb[p] = d->pair_b[p] < S ? new_index[d->pair_b[p]] : d->pair_b[p];
        pu[p] = p;
    }
    // validity tables: pairs stably sorted by class, refs (>= 0 robot shape in frame order, < 0 world ~ref)
    std::vector<int> vcls(P), vorder(P);
    auto kind_of = [&](int ref) { return ref >= 0 ? rs_kind[ref] : ws_kind[~ref]; };
    auto core_of = [&](int ref) -> const double* { return ref >= 0 ? &rs_core[6 * ref] : &ws_core[18 * (~ref) + 12]; };
    std::vector<int> refA(P), refB(P);
    for (int p = 0; p < P; ++p) {
        refA[p] = pa[p];
        refB[p] = pb[p] < S ? pb[p] : ~(pb[p] - S);
        const int ka = kind_of(refA[p]), kb = kind_of(refB[p]);
        const bool a_ps = (ka == K_POINT || ka == K_SEG), b_ps = (kb == K_POINT || kb == K_SEG);
        if (kb == K_PLANE) vcls[p] = 0;
        else if (ka == K_HULL || kb == K_HULL) vcls[p] = 2;           // hulls have no closed forms: GJK, also against a point
        else if ((a_ps && b_ps) || ka == K_POINT || kb == K_POINT) vcls[p] = 1;
        else vcls[p] = 2;
    }
    int n_plane = 0, n_closed = 0, cur = 0;
    for (int c = 0; c < 3; ++c)
        for (int p = 0; p < P; ++p)
            if (vcls[p] == c) { vorder[cur++] = p; if (c == 0) ++n_plane; if (c == 1) ++n_closed; }
    std::vector<int> vp_tab(4 * (size_t)P), vp_canon(2 * (size_t)P);
    std::vector<double> vp_cst(4 * (size_t)P), ws_center(3 * (size_t)W);
    bool margins_zero = true;
    std::vector<double> gjk_margins;
    bool gjk_any_hull = false;
    for (int i = 0; i < P; ++i) {
        const int p = vorder[i];
        const int ka = kind_of(refA[p]), kb = kind_of(refB[p]);
        vp_tab[4 * i] = refA[p]; vp_tab[4 * i + 1] = refB[p]; vp_tab[4 * i + 2] = vcls[p]; vp_tab[4 * i + 3] = p;
        const bool swap = ka > kb;
        vp_canon[2 * i] = swap ? refB[p] : refA[p];
        vp_canon[2 * i + 1] = swap ? refA[p] : refB[p];
        const double* ca = core_of(refA[p]);
        const double* cb = core_of(refB[p]);
        vp_cst[4 * i] = ca[4]; vp_cst[4 * i + 1] = cb[4];
        vp_cst[4 * i + 2] = host_bound_radius(ka, ca);
        vp_cst[4 * i + 3] = host_bound_radius(kb, cb);
        {
            const int k0 = ka < kb ? ka : kb, k1 = ka < kb ? kb : ka;          // canonical order
            const bool closed = k1 != K_HULL && (k0 == K_POINT || ((k0 == K_POINT || k0 == K_SEG) && (k1 == K_POINT || k1 == K_SEG)));
            if (k1 != K_PLANE && !closed && (ca[4] != 0.0 || cb[4] != 0.0)) margins_zero = false;
            if (k1 != K_PLANE && !closed) { gjk_margins.push_back(ca[4]); gjk_margins.push_back(cb[4]); if (k1 == K_HULL) gjk_any_hull = true; }
        }
    }
    for (int w = 0; w < W; ++w) { ws_center[3 * w] = ws_core[18 * w]; ws_center[3 * w + 1] = ws_core[18 * w + 1]; ws_center[3 * w + 2] = ws_core[18 * w + 2]; }
    std::vector<unsigned> frame_mask(J > 0 ? J : 1, 0u), rs_mask(S > 0 ? S : 1, 0u);
    for (int k = 0; k < J; ++k) frame_mask[k] = (d->joint_parent[k] >= 0 ? frame_mask[d->joint_parent[k]] : 0u) | (1u << k);
    for (int i = 0; i < S; ++i) { const int f = d->rshape_frame[order[i]]; rs_mask[i] = f >= 0 ? frame_mask[f] : 0u; }
    // broadphase order: category-major (0 plane, 1 robot-robot, 2 robot-world, 3 robot-world box), then by pair
    std::vector<int> bq_tab(4 * (size_t)(P > 0 ? P : 1), 0);
    {
        int cur_b = 0;
        for (int cat = 0; cat < 4; ++cat)
            for (int i = 0; i < P; ++i) {
                const int p = vorder[i];
                int c;
                if (refB[p] >= 0) c = 1;
                else if (ws_kind[~refB[p]] == K_PLANE) c = 0;
                else if (ws_kind[~refB[p]] == K_BOX) c = 3;
                else c = 2;
                if (c != cat) continue;
                bq_tab[4 * cur_b] = 3 * refA[p];
                bq_tab[4 * cur_b + 1] = refB[p] >= 0 ? 3 * refB[p] : ~refB[p];
                bq_tab[4 * cur_b + 2] = i;
                bq_tab[4 * cur_b + 3] = cat;
                ++cur_b;
            }
    }
    // static reach culling: the centre of robot shape a never leaves the ball of radius reach_a around the base origin
    // (sum of the joint offsets on its path + its local offset; unbounded when a prismatic joint is on the path), so a world
    // shape farther than that from the base, radii included, can never be a candidate.  Rigorous by the triangle inequality.
    std::vector<double> bq_static((size_t)(P > 0 ? P : 1), -INFINITY);
    {
        std::vector<double> reach(S > 0 ? S : 1, 0.0);
        for (int i = 0; i < S; ++i) {
            const int f = d->rshape_frame[order[i]];
            double r = 0.0;
            bool unbounded = false;
            if (f >= 0)
                for (int k = 0; k < J; ++k)
                    if ((frame_mask[f] >> k) & 1u) {
                        if (d->joint_type[k] == NBK_PRISMATIC) unbounded = true;
                        r += std::sqrt(d->joint_trans[3 * k] * d->joint_trans[3 * k] + d->joint_trans[3 * k + 1] * d->joint_trans[3 * k + 1] +
                                       d->joint_trans[3 * k + 2] * d->joint_trans[3 * k + 2]);
                    }
            const double lx = rs_local[12 * i + 3], ly = rs_local[12 * i + 7], lz = rs_local[12 * i + 11];
            r += std::sqrt(lx * lx + ly * ly + lz * lz);
            reach[i] = unbounded ? INFINITY : r * (1.0 + 1e-12) + 1e-12;
        }
        const double b0[3] = {d->base_pose[3], d->base_pose[7], d->base_pose[11]};
        for (int j = 0; j < P; ++j) {
            const int cat = bq_tab[4 * j + 3];
            if (cat == 1) continue;
            const int a = bq_tab[4 * j] / 3, w = bq_tab[4 * j + 1], i = bq_tab[4 * j + 2];
            const double* wc = &ws_core[18 * (size_t)w];
            if (!(reach[a] < INFINITY)) continue;
            if (cat == 0) {
                const double hb = wc[9] * (b0[0] - wc[0]) + wc[10] * (b0[1] - wc[1]) + wc[11] * (b0[2] - wc[2]);
                bq_static[j] = hb - reach[a] - vp_cst[4 * i + 2];
            } else {
                const double dx = wc[0] - b0[0], dy = wc[1] - b0[1], dz = wc[2] - b0[2];
                bq_static[j] = std::sqrt(dx * dx + dy * dy + dz * dz) - reach[a] - (vp_cst[4 * i + 2] + vp_cst[4 * i + 3]);
            }
        }
    }
    if (3 * S >= 65536 || W >= 65536) return NBK_ERR_UNSUPPORTED;
    // the LDS broadphase (robots with more than 16 primitives) keeps the pair constants and world cores in LDS; robots the
    // register broadphases serve do not need it, however many world shapes there are
    const bool lds_broad_ok = (size_t)(d->n_q + 12 * slots + 3 * S) * 64 * sizeof(double) + (4 * (size_t)P + 18 * (size_t)W) * sizeof(double) + BQ_CAP * 4 <= 160 * 1024;
    if (!lds_broad_ok && S > 16) return NBK_ERR_UNSUPPORTED;
    if (P >= (1 << 20)) return NBK_ERR_UNSUPPORTED;
    // LDS budget: q rows + shape rows + saved frames, 512 B each (+ queue and flags of the validity path);
    // the raw q slab reuses the shape area
    const size_t lds_bytes = (size_t)(d->n_q + (rows > d->n_q ? rows : d->n_q) + 12 * slots) * 64 * sizeof(double) + VALIDITY_LDS_EXTRA;
    // robots whose primitives do not fit the LDS-parked layout (some 25+ shapes) keep validity and edges, through the
    // broadphase + narrowphase kernels at every batch size; the per-pair distance entry points report UNSUPPORTED for them
    const bool parked_ok = lds_bytes <= 160 * 1024;
    if (S <= 16 && (size_t)d->n_q * 64 * sizeof(double) + 12 * (size_t)slots * 64 * sizeof(float) + 4096 > 160 * 1024) return NBK_ERR_UNSUPPORTED;
    if (P >= (1 << 26)) return NBK_ERR_UNSUPPORTED;

    Blob B;
    nbk_model* M = new nbk_model();
    memset(&M->d, 0, sizeof(M->d));
    struct Off { size_t jt, jq, jl, js, jb, jr, jtr, jsl, jax, bp, rk, rr, rl, rc, wk, wc, pa, pb, pu, vt, vc, vk, wz, rm, bt, rf, vi, ft, bs, vcl; } o;
    o.jt = B.add(d->joint_type, sizeof(int) * J);
    o.jq = B.add(d->joint_qidx, sizeof(int) * J);
    o.jl = B.add(load.data(), sizeof(int) * J);
    o.js = B.add(save.data(), sizeof(int) * J);
    o.jb = B.add(begin.data(), sizeof(int) * (J + 2));
    // exact zeros of the joint tables are stored as +0 (so that a literal 0.0 in the axis-aligned fast paths is the same operand),
    // and every joint is classified: which coordinate axis of the joint frame it turns about, if any
    std::vector<double> jrot(d->joint_rot, d->joint_rot + 27 * (size_t)J), jtrans(d->joint_trans, d->joint_trans + 3 * (size_t)J);
    for (double& v : jrot) v += 0.0;
    for (double& v : jtrans) v += 0.0;
    std::vector<int> joint_kind(J > 0 ? J : 1, JK_GENERIC);
    for (int k = 0; k < J; ++k) {
        const double* Mk = &jrot[27 * (size_t)k];
        if (d->joint_type[k] == NBK_PRISMATIC) {
            bool zero = true;
            for (int e = 9; e < 27; ++e) zero = zero && Mk[e] == 0.0;
            if (!zero) { delete M; snprintf(g_err, sizeof(g_err), "prismatic joint %d: M1 / M2 of joint_rot must be zero", k); return NBK_ERR_INVALID; }
            joint_kind[k] = JK_PRISMATIC;
            continue;
        }
        for (int kz = 0; kz < 3; ++kz) {
            const int u = (kz + 1) % 3, v = (kz + 2) % 3;
            bool ok = d->joint_slide[3 * k] == 0.0 && d->joint_slide[3 * k + 1] == 0.0 && d->joint_slide[3 * k + 2] == 0.0;
            for (int r = 0; r < 3; ++r)
                ok = ok && Mk[9 + 3 * r + kz] == 0.0 && Mk[18 + 3 * r + kz] == 0.0 && Mk[3 * r + u] == 0.0 && Mk[3 * r + v] == 0.0;
            if (ok) { joint_kind[k] = kz; break; }
        }
    }
    o.jr = B.add(jrot.data(), sizeof(double) * 27 * J);
    const size_t o_jk = B.add(joint_kind.data(), sizeof(int) * J);
    o.jtr = B.add(jtrans.data(), sizeof(double) * 3 * J);
    o.jsl = B.add(d->joint_slide, sizeof(double) * 3 * J);
    std::vector<double> joint_pk(22 * (size_t)(J > 0 ? J : 1), 0.0);
    for (int k = 0; k < J; ++k) {
        const double* Mk = &jrot[27 * (size_t)k];
        const int kz = joint_kind[k] <= 2 ? joint_kind[k] : 0;
        const int u = (kz + 1) % 3, v = (kz + 2) % 3;
        double* jp = &joint_pk[22 * (size_t)k];
        for (int r = 0; r < 3; ++r) jp[18 + r] = d->joint_axis[3 * k + r];
        for (int r = 0; r < 3; ++r) { jp[2 * r] = Mk[18 + 3 * r + u]; jp[2 * r + 1] = Mk[18 + 3 * r + v]; jp[6 + 2 * r] = Mk[9 + 3 * r + u]; jp[6 + 2 * r + 1] = Mk[9 + 3 * r + v]; jp[12 + r] = Mk[3 * r + kz]; jp[15 + r] = jtrans[3 * k + r]; }
    }
    const size_t o_jpk = B.add(joint_pk.data(), sizeof(double) * joint_pk.size());
    M->h_joint_kind.assign(joint_kind.begin(), joint_kind.end());
    o.jax = B.add(d->joint_axis, sizeof(double) * 3 * J);
    o.bp = B.add(d->base_pose, sizeof(double) * 12);
    o.rk = B.add(rs_kind.data(), sizeof(int) * S);
    o.rr = B.add(rs_row.data(), sizeof(int) * S);
    o.rl = B.add(rs_local.data(), sizeof(double) * 12 * S);
    o.rc = B.add(rs_core.data(), sizeof(double) * 6 * S);
    o.wk = B.add(ws_kind.data(), sizeof(int) * W);
    o.wc = B.add(ws_core.data(), sizeof(double) * 18 * W);
    o.pa = B.add(pa.data(), sizeof(int) * P);
    o.pb = B.add(pb.data(), sizeof(int) * P);
    o.pu = B.add(pu.data(), sizeof(int) * P);
    o.vt = B.add(vp_tab.data(), sizeof(int) * 4 * P);
    o.vc = B.add(vp_canon.data(), sizeof(int) * 2 * P);
    o.vk = B.add(vp_cst.data(), sizeof(double) * 4 * P);
    o.wz = B.add(ws_center.data(), sizeof(double) * 3 * W);
    o.rm = B.add(rs_mask.data(), sizeof(unsigned) * S);
    std::vector<int> vp_info(4 * (size_t)(P > 0 ? P : 1), 0);
    for (int i = 0; i < P; ++i) {
        const int ra = vp_canon[2 * i], rb = vp_canon[2 * i + 1];
        vp_info[4 * i] = ra; vp_info[4 * i + 1] = rb;
        vp_info[4 * i + 2] = ra >= 0 ? (int)rs_mask[ra] : 0;
        vp_info[4 * i + 3] = rb >= 0 ? (int)rs_mask[rb] : 0;
    }
    o.vi = B.add(vp_info.data(), sizeof(int) * 4 * P);
    std::vector<int> vp_cls((size_t)(P > 0 ? P : 1), 3);
    int cls_count[4] = {0, 0, 0, 0};
    for (int i = 0; i < P; ++i) {
        const int ra = vp_canon[2 * i], rb = vp_canon[2 * i + 1];
        const int ka = ra >= 0 ? rs_kind[ra] : ws_kind[~ra], kb = rb >= 0 ? rs_kind[rb] : ws_kind[~rb];
        vp_cls[i] = (ka == K_BOX && kb == K_BOX) ? 0 : ((ka == K_BOX && kb == K_CYL) ? 1 : ((ka == K_CYL && kb == K_CYL) ? 2 : 3));
        cls_count[vp_cls[i]] += 1;
    }
    // sub-queues per class in proportion to its pairs (at least one for a class that has pairs)
    int cls_groups[4] = {0, 0, 0, 0}, cls_base[4] = {0, 0, 0, 0};
    {
        int used = 0, nonempty = 0;
        for (int c = 0; c < 4; ++c) if (cls_count[c] > 0) ++nonempty;
        const int spare = NSUB - nonempty;
        for (int c = 0; c < 4; ++c)
            if (cls_count[c] > 0) { cls_groups[c] = 1 + (int)((long long)spare * cls_count[c] / (P > 0 ? P : 1)); used += cls_groups[c]; }
        // hand what rounding left over to the largest class
        int big = 0;
        for (int c = 1; c < 4; ++c) if (cls_count[c] > cls_count[big]) big = c;
        if (P > 0) cls_groups[big] += NSUB - used;
        for (int c = 1; c < 4; ++c) cls_base[c] = cls_base[c - 1] + cls_groups[c - 1];
    }
    o.vcl = B.add(vp_cls.data(), sizeof(int) * P);
    // float32 tables + error slack of the conservative broadphase.  Position error of a float32 chain sweep is below
    // (joints + 2) * 16 ulp(float) * reach; the slack is 50x that, never below 1e-4 of the reach.
    // local bounding box of every hull (centre, half extents rounded outwards): the hull midphase culls against it
    std::vector<double> hull_obb(6 * (size_t)(H > 0 ? H : 1), 0.0);
    for (int h = 0; h < H; ++h) {
        const double* v = d->hull_verts + 3 * (size_t)d->hull_vert_begin[h];
        const int n = d->hull_vert_begin[h + 1] - d->hull_vert_begin[h];
        double lo[3] = {v[0], v[1], v[2]}, hi[3] = {v[0], v[1], v[2]};
        for (int k = 1; k < n; ++k)
            for (int j = 0; j < 3; ++j) { lo[j] = std::min(lo[j], v[3 * k + j]); hi[j] = std::max(hi[j], v[3 * k + j]); }
        for (int j = 0; j < 3; ++j) { hull_obb[6 * h + j] = 0.5 * (lo[j] + hi[j]); hull_obb[6 * h + 3 + j] = 0.5 * (hi[j] - lo[j]) * (1.0 + 1e-12) + 1e-300; }
    }
    std::vector<float> ftab;
    int f_trans, f_slide, f_base, f_tl, f_wc, f_wobb = 0, f_pk = 0, f_meta = 0, f_chain = 0;
    double reach = 0.0;
    {
        for (int k = 0; k < J; ++k) for (int e = 0; e < 27; ++e) ftab.push_back((float)d->joint_rot[27 * k + e]);
        f_trans = (int)ftab.size();
        for (int k = 0; k < J; ++k) {
            double n2 = 0.0;
            for (int e = 0; e < 3; ++e) { ftab.push_back((float)d->joint_trans[3 * k + e]); n2 += d->joint_trans[3 * k + e] * d->joint_trans[3 * k + e]; }
            reach += std::sqrt(n2);
        }
        f_slide = (int)ftab.size();
        for (int k = 0; k < J; ++k) for (int e = 0; e < 3; ++e) ftab.push_back((float)d->joint_slide[3 * k + e]);
        f_base = (int)ftab.size();
        {
            double n2 = 0.0;
            for (int e = 0; e < 12; ++e) ftab.push_back((float)d->base_pose[e]);
            for (int i = 0; i < 3; ++i) n2 += d->base_pose[4 * i + 3] * d->base_pose[4 * i + 3];
            reach += std::sqrt(n2);
        }
        f_tl = (int)ftab.size();
        double lmax = 0.0;
        for (int i = 0; i < S; ++i) {
            double n2 = 0.0;
            for (int r = 0; r < 3; ++r) { const double v = rs_local[12 * i + 4 * r + 3]; ftab.push_back((float)v); n2 += v * v; }
            if (std::sqrt(n2) > lmax) lmax = std::sqrt(n2);
        }
        reach += lmax;
        f_wc = (int)ftab.size();
        for (int w = 0; w < W; ++w) {
            double n2 = 0.0;
            for (int e = 0; e < 18; ++e) ftab.push_back((float)ws_core[18 * w + e]);
            for (int e = 0; e < 3; ++e) n2 += ws_core[18 * w + e] * ws_core[18 * w + e];
            if (std::sqrt(n2) > reach) reach = std::sqrt(n2);          // world coordinates enter the differences too
        }
        f_wobb = (int)ftab.size();
        for (int w = 0; w < W; ++w)
            for (int e = 0; e < 6; ++e) {
                const double v = ws_hull[w] >= 0 ? hull_obb[6 * (size_t)ws_hull[w] + e] : 0.0;
                ftab.push_back(e < 3 ? (float)v : (float)v * (1.0f + 2.4e-7f));          // half extents rounded up
            }
        // per-joint constants of the packed sweep (k_broad_f32): for a joint about coordinate axis KZ of its frame (U, V = the two
        // other axes) the pairs (M2[r][U], M2[r][V]) and (M1[r][U], M1[r][V]), r = 0..2, column KZ of M0, the offset translation
        while (ftab.size() % 4 != 0) ftab.push_back(0.0f);
        f_pk = (int)ftab.size();
        for (int k = 0; k < (J > 8 ? J : 8); ++k) {
            if (k >= J) { for (int e = 0; e < 20; ++e) ftab.push_back(0.0f); continue; }      // (the chain sweep prefetches entry k + 1 <= 7)
            const double* M = d->joint_rot + 27 * (size_t)k;
            const int kz = joint_kind[k] <= 2 ? joint_kind[k] : 0;
            const int u = (kz + 1) % 3, v = (kz + 2) % 3;
            for (int r = 0; r < 3; ++r) { ftab.push_back((float)M[18 + 3 * r + u]); ftab.push_back((float)M[18 + 3 * r + v]); }
            for (int r = 0; r < 3; ++r) { ftab.push_back((float)M[9 + 3 * r + u]); ftab.push_back((float)M[9 + 3 * r + v]); }
            for (int r = 0; r < 3; ++r) ftab.push_back((float)M[3 * r + kz]);
            ftab.push_back(0.0f);
            for (int e = 0; e < 3; ++e) ftab.push_back((float)d->joint_trans[3 * k + e]);
            ftab.push_back(0.0f);
        }
        f_meta = (int)ftab.size();
        for (int k = 0; k < 8; ++k) {
            const unsigned v = k < J ? ((unsigned)joint_kind[k] | ((unsigned)d->joint_qidx[k] << 8)) : 0u;
            float fv; memcpy(&fv, &v, 4);
            ftab.push_back(fv);
        }
        f_chain = (J >= 1 && J <= 8 && S <= 16) ? 1 : 0;
        for (int k = 0; k < J && f_chain; ++k) if (load[k] != (k == 0 ? -1 : -2) || save[k] != -1) f_chain = 0;
        if (ftab.empty()) ftab.push_back(0.0f);
    }
    o.ft = B.add(ftab.data(), sizeof(float) * ftab.size());
    std::vector<int> rs_frame_v(S > 0 ? S : 1, -1);
    for (int i = 0; i < S; ++i) rs_frame_v[i] = d->rshape_frame[order[i]];
    o.rf = B.add(rs_frame_v.data(), sizeof(int) * S);
    o.bt = B.add(bq_tab.data(), sizeof(int) * 4 * P);
    o.bs = B.add(bq_static.data(), sizeof(double) * P);
    // radius of a ball around each shape's centre that lies inside the shape (the float32 broadphase certifies a collision when two
    // such balls overlap): margin + the smallest half extent of the core; hulls: the smallest face offset (0 without planes)
    std::vector<double> rs_in(S > 0 ? S : 1, 0.0), ws_in(W > 0 ? W : 1, 0.0);
    {
        auto inscribed = [&](int kind, const double* cc, int hull) {
            double r = 0.0;
            if (kind == K_BOX) r = std::min(cc[0], std::min(cc[1], cc[2]));
            else if (kind == K_CYL) r = std::min(cc[3], cc[0]);
            else if (kind == K_HULL) {
                const int f0 = d->hull_face_begin[hull], f1 = d->hull_face_begin[hull + 1];
                r = f1 > f0 ? INFINITY : 0.0;
                for (int f = f0; f < f1; ++f) r = std::min(r, d->hull_planes[4 * (size_t)f + 3]);
                r *= (1.0 - 1e-9);             // the planes come from a float64 hull computation: stay inside them
            } else if (kind == K_PLANE) return 0.0;
            if (!(r > 0.0)) r = 0.0;
            return r + cc[4];
        };
        for (int i = 0; i < S; ++i) rs_in[i] = inscribed(rs_kind[i], &rs_core[6 * (size_t)i], rs_hull[i]);
        for (int w = 0; w < W; ++w) ws_in[w] = inscribed(ws_kind[w], &ws_core[18 * (size_t)w + 12], ws_hull[w]);
    }
    const size_t o_rin = B.add(rs_in.data(), sizeof(double) * S);
    const size_t o_win = B.add(ws_in.data(), sizeof(double) * W);
    // hull vertices, each hull's list preceded by its local bounding box (centre, half extents): 6 + 3 n doubles per hull
    std::vector<double> hull_blob;
    std::vector<size_t> hull_off(H > 0 ? H : 1, 0);           // offset (in doubles) of hull h's first vertex inside hull_blob
    for (int h = 0; h < H; ++h) {
        const double* v = d->hull_verts + 3 * (size_t)d->hull_vert_begin[h];
        const int n = d->hull_vert_begin[h + 1] - d->hull_vert_begin[h];
        hull_blob.insert(hull_blob.end(), &hull_obb[6 * h], &hull_obb[6 * h] + 6);
        hull_off[h] = hull_blob.size();
        hull_blob.insert(hull_blob.end(), v, v + 3 * (size_t)n);
    }
    const size_t o_hv = B.add(hull_blob.data(), sizeof(double) * hull_blob.size());
    const size_t o_hp = B.add(d->hull_planes, sizeof(double) * 4 * (size_t)(H > 0 ? d->hull_face_begin[H] : 0));
    B.bytes.resize((B.bytes.size() + 255) & ~size_t(255));

    void* dev = nullptr;
    hipError_t e = hipMalloc(&dev, B.bytes.size());
    if (e != hipSuccess) { delete M; hip_fail(e, "hipMalloc(model)"); return NBK_ERR_ALLOC; }
    // hull shapes: their 24-byte h[] slots in the shape tables hold the device addresses of the hull's vertices / planes
    {
        auto patch = [&](size_t slot, int h) {
            HullRef r;
            r.hv = reinterpret_cast<const double*>(static_cast<const char*>(dev) + o_hv) + hull_off[h];
            r.hp = reinterpret_cast<const double*>(static_cast<const char*>(dev) + o_hp) + 4 * (size_t)d->hull_face_begin[h];
            r.hn = d->hull_vert_begin[h + 1] - d->hull_vert_begin[h];
            r.hf = d->hull_face_begin[h + 1] - d->hull_face_begin[h];
            static_assert(sizeof(HullRef) == 24, "HullRef must overlay h[3]");
            memcpy(B.bytes.data() + slot, &r, sizeof(r));
        };
        for (int i = 0; i < S; ++i) if (rs_hull[i] >= 0) patch(o.rc + sizeof(double) * 6 * (size_t)i, rs_hull[i]);
        for (int w = 0; w < W; ++w) if (ws_hull[w] >= 0) patch(o.wc + sizeof(double) * (18 * (size_t)w + 12), ws_hull[w]);
    }
    e = hipMemcpy(dev, B.bytes.data(), B.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(dev); delete M; return hip_fail(e, "hipMemcpy(model)"); }
    const char* base = static_cast<const char*>(dev);
    DevModel& m = M->d;
    m.n_q = d->n_q; m.n_joints = J; m.n_rshapes = S; m.n_wshapes = W; m.n_pairs = P;
    m.shape_rows = rows > d->n_q ? rows : d->n_q;
    m.frame_slots = slots;
    m.joint_type = reinterpret_cast<const int*>(base + o.jt);
    m.joint_qidx = reinterpret_cast<const int*>(base + o.jq);
    m.joint_load = reinterpret_cast<const int*>(base + o.jl);
    m.joint_save = reinterpret_cast<const int*>(base + o.js);
    m.joint_shape_begin = reinterpret_cast<const int*>(base + o.jb);
    m.joint_rot = reinterpret_cast<const double*>(base + o.jr);
    m.joint_kind = reinterpret_cast<const int*>(base + o_jk);
    m.joint_trans = reinterpret_cast<const double*>(base + o.jtr);
    m.joint_slide = reinterpret_cast<const double*>(base + o.jsl);
    m.joint_axis = reinterpret_cast<const double*>(base + o.jax);
    m.joint_pk = reinterpret_cast<const double*>(base + o_jpk);
    m.base_pose = reinterpret_cast<const double*>(base + o.bp);
    m.rs_kind = reinterpret_cast<const int*>(base + o.rk);
    m.rs_row = reinterpret_cast<const int*>(base + o.rr);
    m.rs_local = reinterpret_cast<const double*>(base + o.rl);
    m.rs_core = reinterpret_cast<const double*>(base + o.rc);
    m.ws_kind = reinterpret_cast<const int*>(base + o.wk);
    m.ws_core = reinterpret_cast<const double*>(base + o.wc);
    m.hull_blob = reinterpret_cast<const double*>(base + o_hv);
    m.hull_blob_n = (int)hull_blob.size();
    m.pair_a = reinterpret_cast<const int*>(base + o.pa);
    m.pair_b = reinterpret_cast<const int*>(base + o.pb);
    m.pair_user = reinterpret_cast<const int*>(base + o.pu);
    m.vp_tab = reinterpret_cast<const int*>(base + o.vt);
    m.vp_canon = reinterpret_cast<const int*>(base + o.vc);
    m.vp_cst = reinterpret_cast<const double*>(base + o.vk);
    m.ws_center = reinterpret_cast<const double*>(base + o.wz);
    m.n_plane_pairs = n_plane; m.n_closed_pairs = n_closed;
    m.rs_mask = reinterpret_cast<const unsigned*>(base + o.rm);
    m.vp_info = reinterpret_cast<const int4*>(base + o.vi);
    m.vp_cls = reinterpret_cast<const int*>(base + o.vcl);
    for (int c = 0; c < 4; ++c) { m.cls_base[c] = cls_base[c]; m.cls_groups[c] = cls_groups[c] > 0 ? cls_groups[c] : 1; }
    m.f_tab = reinterpret_cast<const float*>(base + o.ft);
    m.f_trans = f_trans; m.f_slide = f_slide; m.f_base = f_base; m.f_tl = f_tl; m.f_wc = f_wc; m.f_wobb = f_wobb; m.f_pk = f_pk; m.f_meta = f_meta; m.f_chain = f_chain;
    // relative slack: 50 x the float32 error bound (joints + 2) * 16 ulp of a chain sweep; the kernel multiplies it by the
    // larger of the static reach and the configuration's own largest coordinate (prismatic travel is unbounded here)
    {
        const double rel = 50.0 * (J + 2) * 16.0 * 5.96e-8;
        m.f_eps = (float)(rel > 1e-4 ? rel : 1e-4);
        m.f_reach = (float)(reach > 1e-3 ? reach : 1e-3);
        m.f_e2max = 2.0f * m.f_reach * (m.f_eps + 2.4e-7f * 64.0f) * (1.0f + 1e-6f);
    }
    m.rs_frame = reinterpret_cast<const int*>(base + o.rf);
    m.bq_tab = reinterpret_cast<const int*>(base + o.bt);
    m.bq_static = reinterpret_cast<const double*>(base + o.bs);
    m.rs_in = reinterpret_cast<const double*>(base + o_rin);
    m.ws_in = reinterpret_cast<const double*>(base + o_win);
    for (int c = 0; c < 4; ++c) { m.bq_count[c] = 0; }
    for (int i = 0; i < P; ++i) m.bq_count[bq_tab[4 * i + 3]]++;
#ifdef NBK_ABLATE_BUILD
    { const char* ab = getenv("NBK_ABLATE"); m.dbg = ab ? atoi(ab) : 0; }
#else
    m.dbg = 0;
#endif
    M->blob = dev;
    M->scalar_q = nullptr; M->scalar_out = nullptr; M->scalar_stream = nullptr;
    M->blob_bytes = B.bytes.size();
    M->n_pairs = P; M->n_q = d->n_q; M->n_joints = J;
    M->h_joint_qidx.assign(d->joint_qidx, d->joint_qidx + J);
    M->h_joint_type.assign(d->joint_type, d->joint_type + J);
    M->margins_zero = margins_zero;
    for (int c = 0; c < 4; ++c) M->cls_count[c] = cls_count[c];
    M->h_static = bq_static; M->h_static.resize(P > 0 ? P : 0);
    M->h_m0.resize(P); M->h_m1.resize(P); M->h_cat.resize(P); M->h_cls.resize(P);
    for (int j = 0; j < P; ++j) {
        const int i = bq_tab[4 * j + 2];
        M->h_m0[j] = vp_cst[4 * (size_t)i]; M->h_m1[j] = vp_cst[4 * (size_t)i + 1];
        M->h_cat[j] = bq_tab[4 * j + 3]; M->h_cls[j] = vp_cls[i];
    }
    M->gjk_margins = gjk_margins;
    M->gjk_any_hull = gjk_any_hull;
    M->world_hulls = world_hulls;
    M->lds_broad_ok = lds_broad_ok;
    M->parked_ok = parked_ok;
    (void)hipGetDevice(&M->device);
    *out = M;
    return NBK_OK;
}

void nbk_model_destroy(nbk_model* m) {
    if (m == nullptr) return;
    if (m->blob) (void)hipFree(m->blob);
    for (StreamWs* w : m->wss) {
        if (w->ws) (void)hipFree(w->ws);
        if (w->ews) (void)hipFree(w->ews);
        if (w->stats) (void)hipHostFree(w->stats);
        if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
        if (w->ev_join) (void)hipEventDestroy(w->ev_join);
        if (w->aux_stream) (void)hipStreamDestroy(w->aux_stream);
        delete w;
    }
    if (m->scalar_q) (void)hipHostFree(m->scalar_q);
    if (m->scalar_out) (void)hipHostFree(m->scalar_out);
    if (m->scalar_stream) (void)hipStreamDestroy(m->scalar_stream);
    delete m;
}

// diagnostic (not part of include/nbk.h): cycles per phase of k_narrow accumulated since the last reset, out[16]
extern "C" int32_t nbk_debug_narrow_profile(unsigned long long* out, int32_t reset) {
    if (out != nullptr) NBK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(nbk::g_narrow_prof), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; z[13] = ~0ull; NBK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(nbk::g_narrow_prof), z, sizeof(z))); }
    return NBK_OK;
}

// diagnostic (not part of include/nbk.h): cycles per phase of k_broad_f32 (builds with -DNBK_BF32_STAMP), out[8]
extern "C" int32_t nbk_debug_broad_profile(unsigned long long* out, int32_t reset) {
#ifdef NBK_BF32_STAMP
    static std::vector<unsigned long long> h(16384 * 8);
    if (out != nullptr) {
        NBK_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(nbk::g_broad_prof), sizeof(unsigned long long) * h.size()));
        for (int e = 0; e < 8; ++e) { out[e] = 0; for (size_t b = 0; b < 16384; ++b) out[e] += h[8 * b + e]; }
    }
    if (reset) { std::fill(h.begin(), h.end(), 0ull); NBK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(nbk::g_broad_prof), h.data(), sizeof(unsigned long long) * h.size())); }
    return NBK_OK;
#else
    (void)out; (void)reset;
    return NBK_ERR_UNSUPPORTED;
#endif
}

int32_t nbk_model_num_pairs(const nbk_model* m) { return m ? m->n_pairs : NBK_ERR_INVALID; }

static int make_path(const nbk_model* m, const int32_t* path, int32_t path_len, const double* local, PathArg& pa) {
    if (path_len < 0 || path_len > NBK_MAX_JOINTS || (path_len > 0 && path == nullptr) || local == nullptr) return NBK_ERR_INVALID;
    pa.len = path_len;
    for (int i = 0; i < path_len; ++i) {
        if (path[i] < 0 || path[i] >= m->n_joints) return NBK_ERR_INVALID;
        pa.idx[i] = path[i];
    }
    pa.revolute = 0u; pa.covered = 0u;
    for (int i = 0; i < path_len; ++i) {
        pa.col[i] = m->h_joint_qidx[path[i]];
        if (m->h_joint_type[path[i]] == NBK_REVOLUTE) pa.revolute |= 1u << i;
        if (pa.col[i] >= 0 && pa.col[i] < 32) pa.covered |= 1u << pa.col[i];
    }
    for (int i = path_len; i < NBK_MAX_JOINTS; ++i) { pa.idx[i] = 0; pa.col[i] = 0; }
    for (int i = 0; i < NBK_MAX_JOINTS; ++i) pa.kind[i] = (unsigned char)(i < path_len ? m->h_joint_kind[path[i]] : 0);
    memcpy(pa.local, local, sizeof(double) * 12);
    return NBK_OK;
}

// ---- tuning / diagnostic switches: process-wide, seeded ONCE from the environment when the library is loaded -------
// (no getenv on any call path).  nbk_debug_set_option changes them at run time (tests, tools); none of them changes a result.
struct Options {
    long long two_kernel_min_b;     // NBK_TWO_KERNEL_MIN_B: batches below this size run the fused kernel k_validity.  The broadphase +
                                    // narrowphase pair measured faster at every size (0.040 vs 0.050 ms for 64 configurations), so: 1
    long long edge_batch_min_e;     // NBK_EDGE_BATCH_MIN_E: edge batches below this size run one wave per edge (k_edges); default 1
    long long no_reg_broad;         // NBK_NO_REG_BROAD: the LDS broadphase k_broad instead of the register broadphases
    long long f64_broad;            // NBK_F64_BROAD: the float64 register broadphase instead of the conservative float32 one
    long long jac_two_sweep;        // NBK_JAC_TWO_SWEEP: the general Jacobian kernel also for short paths
    long long closest_brute;        // NBK_CLOSEST_BRUTE: every pair instead of branch-and-bound
    long long narrow_parts_max;     // NBK_NARROW_PARTS_MAX: cap of the narrowphase workgroups per sub-queue
    long long pipeline_tiles;       // NBK_PIPELINE_TILES: batches of >= 2 x 2^20 configurations run their tiles alternately on two streams (default 1)
    long long pipe_tile;            // NBK_PIPE_TILE: configurations per tile of a pipelined batch (default 2^20)
    long long queue_budget;         // NBK_QUEUE_BUDGET: bytes the item queues of one tile may take (default 1 GiB); tests shrink it to force
                                    // the overflow path (k_validity_redo)
    long long fk_lds_q;             // NBK_FK_LDS_Q=1: k_fk stages q in LDS also for n_q <= 8 (A/B switch)
};
static long long env_ll(const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; }
static Options g_opt = {env_ll("NBK_TWO_KERNEL_MIN_B", 1), env_ll("NBK_EDGE_BATCH_MIN_E", 1), env_ll("NBK_NO_REG_BROAD", 0),
                        env_ll("NBK_F64_BROAD", 0), env_ll("NBK_JAC_TWO_SWEEP", 0), env_ll("NBK_CLOSEST_BRUTE", 0), env_ll("NBK_NARROW_PARTS_MAX", 16), env_ll("NBK_PIPELINE_TILES", 1), env_ll("NBK_PIPE_TILE", 1ll << 20), env_ll("NBK_QUEUE_BUDGET", 1ll << 30), env_ll("NBK_FK_LDS_Q", 0)};

// diagnostic (not part of include/nbk.h): set one of the switches above by name; returns NBK_ERR_INVALID for an unknown name
extern "C" int32_t nbk_debug_set_option(const char* name, int64_t value) {
    if (name == nullptr) return NBK_ERR_INVALID;
    struct { const char* n; long long* v; } tab[] = {
        {"two_kernel_min_b", &g_opt.two_kernel_min_b}, {"edge_batch_min_e", &g_opt.edge_batch_min_e}, {"no_reg_broad", &g_opt.no_reg_broad},
        {"f64_broad", &g_opt.f64_broad}, {"jac_two_sweep", &g_opt.jac_two_sweep}, {"closest_brute", &g_opt.closest_brute},
        {"narrow_parts_max", &g_opt.narrow_parts_max}, {"queue_budget", &g_opt.queue_budget}, {"pipeline_tiles", &g_opt.pipeline_tiles}, {"pipe_tile", &g_opt.pipe_tile}, {"fk_lds_q", &g_opt.fk_lds_q}};
    for (auto& t : tab) if (strcmp(t.n, name) == 0) { *t.v = (long long)value; return NBK_OK; }
    return NBK_ERR_INVALID;
}

// every compute entry point: the descriptor's memory lives on the device it was created on
static int32_t check_device(const nbk_model* m) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != m->device) {
        snprintf(g_err, sizeof(g_err), "descriptor belongs to device %d, the current device is %d", m->device, dev);
        return NBK_ERR_INVALID;
    }
    return NBK_OK;
}
#define NBK_DEVICE(m) do { const int32_t d_ = check_device(m); if (d_ != NBK_OK) return d_; } while (0)

static bool stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

static inline unsigned blocks_for(int64_t B) { return (unsigned)((B + WAVE - 1) / WAVE); }

int32_t nbk_fk_batch(const nbk_model* m, const double* q, int64_t B, const int32_t* path, int32_t path_len,
                     const double* local, const double* local_pose, double* T_out, void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || T_out == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    PathArg pa;
    const int st = make_path(m, path, path_len, local, pa);
    if (st != NBK_OK) return st;
    if (B == 0) return NBK_OK;
    const size_t lds = sizeof(double) * WAVE * ((size_t)(m->n_q > 17 ? m->n_q : 17));
    if (m->n_q <= 8 && !g_opt.fk_lds_q) hipLaunchKernelGGL(k_fk<true>, dim3(blocks_for(B)), dim3(WAVE), sizeof(double) * WAVE * 17, (hipStream_t)stream, m->d, pa, q, B, local_pose, T_out);
    else hipLaunchKernelGGL(k_fk<false>, dim3(blocks_for(B)), dim3(WAVE), lds, (hipStream_t)stream, m->d, pa, q, B, local_pose, T_out);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

// ---- frame sets: FK of many frames per configuration --------------------------------------------------------------
struct nbk_frameset {
    int n;
    void* blob;                 // begin[J+2] | out[n] | local[n][12]
    const int* begin;
    const int* out;
    const double* local;
};

int32_t nbk_frameset_create(const nbk_model* m, int32_t n_frames, const int32_t* frame_joint, const double* frame_local,
                            nbk_frameset** out) {
    if (m == nullptr || out == nullptr || n_frames < 1 || n_frames > 4096 || frame_joint == nullptr || frame_local == nullptr) return NBK_ERR_INVALID;
    const int J = m->n_joints;
    for (int f = 0; f < n_frames; ++f) if (frame_joint[f] < -1 || frame_joint[f] >= J) return NBK_ERR_INVALID;
    std::vector<int> order(n_frames), begin(J + 2, 0), outv(n_frames);
    for (int f = 0; f < n_frames; ++f) order[f] = f;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return frame_joint[a] < frame_joint[b]; });
    for (int f = 0; f < n_frames; ++f) begin[frame_joint[f] + 2] += 1;
    for (int k = 1; k < J + 2; ++k) begin[k] += begin[k - 1];
    std::vector<double> local(12 * (size_t)n_frames);
    for (int i = 0; i < n_frames; ++i) { outv[i] = order[i]; memcpy(&local[12 * i], frame_local + 12 * order[i], 12 * sizeof(double)); }
    Blob Bb;
    const size_t ob = Bb.add(begin.data(), sizeof(int) * (J + 2));
    const size_t oo = Bb.add(outv.data(), sizeof(int) * n_frames);
    const size_t ol = Bb.add(local.data(), sizeof(double) * 12 * n_frames);
    nbk_frameset* fs = new nbk_frameset();
    fs->n = n_frames; fs->blob = nullptr;
    hipError_t e = hipMalloc(&fs->blob, Bb.bytes.size());
    if (e != hipSuccess) { delete fs; hip_fail(e, "hipMalloc(frameset)"); return NBK_ERR_ALLOC; }
    e = hipMemcpy(fs->blob, Bb.bytes.data(), Bb.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(fs->blob); delete fs; return hip_fail(e, "hipMemcpy(frameset)"); }
    const char* b = static_cast<const char*>(fs->blob);
    fs->begin = reinterpret_cast<const int*>(b + ob);
    fs->out = reinterpret_cast<const int*>(b + oo);
    fs->local = reinterpret_cast<const double*>(b + ol);
    *out = fs;
    return NBK_OK;
}

void nbk_frameset_destroy(nbk_frameset* fs) {
    if (fs == nullptr) return;
    if (fs->blob) (void)hipFree(fs->blob);
    delete fs;
}

int32_t nbk_fk_frames_batch(const nbk_model* m, const nbk_frameset* fs, const double* q, int64_t B, double* T_out, void* stream) {
    if (m == nullptr || fs == nullptr || B < 0 || (B > 0 && (q == nullptr || T_out == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (B == 0) return NBK_OK;
    const size_t tr = (size_t)(m->n_q > 17 ? m->n_q : 17);
    const size_t lds = sizeof(double) * WAVE * ((size_t)m->n_q + 12 * (size_t)m->d.frame_slots + tr);
    if (lds > 160 * 1024) return NBK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_fk_frames, dim3(blocks_for(B)), dim3(WAVE), lds, (hipStream_t)stream, m->d, q, B, fs->n, fs->begin, fs->out,
                       fs->local, T_out);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

int32_t nbk_knn_prefix(const float* points, int32_t n_points, int32_t dim, int32_t k, int32_t* out_idx, void* stream) {
    if (n_points < 0 || dim < 1 || dim > 64 || k < 1 || k > 64 || (n_points > 0 && (points == nullptr || out_idx == nullptr))) return NBK_ERR_INVALID;
    if (n_points == 0) return NBK_OK;
    hipLaunchKernelGGL(k_knn_prefix, dim3((unsigned)n_points), dim3(WAVE), 0, (hipStream_t)stream, points, n_points, dim, k, out_idx);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

int32_t nbk_jacobian_batch(const nbk_model* m, const double* q, int64_t B, const int32_t* path, int32_t path_len,
                           const double* local, int32_t mode, const double* pose, double* J_out, void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || J_out == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (mode < 0 || mode > 2 || (mode != 0 && pose == nullptr && B > 0)) return NBK_ERR_INVALID;
    PathArg pa;
    const int st = make_path(m, path, path_len, local, pa);
    if (st != NBK_OK) return st;
    if (B == 0) return NBK_OK;
    if (pa.len <= 8 && !g_opt.jac_two_sweep) {
        const int stride = (6 * m->n_q) | 1;
        const size_t lds_q = (size_t)WAVE * (size_t)m->n_q, lds_rows = (size_t)JAC_ROWS * (size_t)stride;     // the rows reuse the q area
        const size_t lds = sizeof(double) * (lds_q > lds_rows ? lds_q : lds_rows);
        hipLaunchKernelGGL(k_jacobian_reg<8>, dim3(blocks_for(B)), dim3(WAVE), lds, (hipStream_t)stream, m->d, pa, q, B, mode | (g_opt.fk_lds_q ? 256 : 0), pose, J_out);
        NBK_HIP(hipGetLastError());
        return NBK_OK;
    }
    const int ncol = 6 * m->n_q;
    const int stride = ncol + 1 + ((ncol + 1) & 1 ? 0 : 1);
    const size_t lds = sizeof(double) * WAVE * ((size_t)m->n_q + (size_t)stride);
    if (lds > 160 * 1024) return NBK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_jacobian, dim3(blocks_for(B)), dim3(WAVE), lds, (hipStream_t)stream, m->d, pa, q, B, mode, pose, J_out);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

int32_t nbk_ik_batch(const nbk_model* m, const double* pose, const double* q0, int64_t B, const int32_t* path, int32_t path_len,
                     const double* local, const double* limits, double tol, int32_t max_iter, int32_t max_failures,
                     double* q_out, uint8_t* success, double* diff_norm, int32_t* iters, void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && (pose == nullptr || q0 == nullptr || q_out == nullptr || success == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (max_iter < 1 || max_failures < 0 || !(tol >= 0.0)) return NBK_ERR_INVALID;
    PathArg pa;
    const int st = make_path(m, path, path_len, local, pa);
    if (st != NBK_OK) return st;
    if (B == 0) return NBK_OK;
    IkArg arg;
    memset(&arg, 0, sizeof(arg));
    arg.tol = tol; arg.max_iter = max_iter; arg.max_failures = max_failures; arg.use_limits = limits != nullptr ? 1 : 0;
    if (limits != nullptr)
        for (int j = 0; j < m->n_q; ++j) { arg.lo[j] = limits[2 * j]; arg.hi[j] = limits[2 * j + 1]; }
    const size_t lds = sizeof(double) * WAVE * ((size_t)m->n_q * 7 + 6 * (size_t)(pa.len > 0 ? pa.len : 1));
    if (lds > 160 * 1024) return NBK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_ik, dim3(blocks_for(B)), dim3(WAVE), lds, (hipStream_t)stream, m->d, pa, arg, pose, q0, B, q_out, success,
                       diff_norm, iters);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

static inline size_t collide_lds(const nbk_model* m) {
    return sizeof(double) * WAVE * ((size_t)m->d.n_q + (size_t)m->d.shape_rows + 12 * (size_t)m->d.frame_slots) + VALIDITY_LDS_EXTRA;
}

// ---- validity: fused kernel for small batches, broadphase + compacted narrowphase for large ones -------
static const size_t WS_MAX_BYTES = size_t(1) << 30;
static const int64_t TILE_MAX = int64_t(1) << 22;               // configurations per queue tile at most (65 536 blocks)
static const size_t WS_FLAGS = (size_t)(TILE_MAX / WAVE);       // one overflow mark per block of a tile
static const size_t WS_COUNTER_SET = NSUB * CNT_STRIDE * 8;    // NSUB counters, one cache line each
static const size_t WS_COUNTERS = 2 * WS_COUNTER_SET;          // two sets (see StreamWs::epoch)
static inline size_t ws_tables(const nbk_model* m) {           // counters | per-call float32 broadphase tables
    return (WS_COUNTERS + 4 * (5 * 256 + 128 + 6 * (size_t)m->d.n_wshapes * 16 + 32 + (size_t)m->d.n_wshapes + 16 + 96 * (size_t)m->d.n_wshapes + 16) + 255) & ~size_t(255);
}
static inline size_t ws_header(const nbk_model* m) { return ws_tables(m) + WS_FLAGS; }      // ... | overflow marks | items follow

static inline size_t broad_lds(const nbk_model* m) {
    const size_t qrows = ((size_t)WAVE * m->d.n_q * 8 >= (size_t)BQ_CAP * 4) ? (size_t)m->d.n_q : ((size_t)BQ_CAP * 4 + WAVE * 8 - 1) / (WAVE * 8);
    return sizeof(double) * (WAVE * (qrows + 12 * (size_t)m->d.frame_slots + 3 * (size_t)m->d.n_rshapes) + 4 * (size_t)m->d.n_pairs + 18 * (size_t)m->d.n_wshapes);
}

// capacity (items) of one sub-queue for a tile of nblk 64-configuration blocks: the blocks that feed it times 64 times
// the pairs of its class, maximised over the classes (all sub-queues get the same stride)
struct PairCounts { int n[4]; };      // pairs per kind class that can produce queue items
static inline PairCounts all_pairs(const nbk_model* m) { PairCounts c; for (int i = 0; i < 4; ++i) c.n[i] = m->cls_count[i]; return c; }
// the pairs the static reach test leaves at this threshold (the very test k_prepare_f32 applies: a pair it drops can never pass
// the bounding-sphere test of any broadphase, so it can never own a queue item)
static inline PairCounts reachable_pairs(const nbk_model* m, double thr) {
    PairCounts c = {{0, 0, 0, 0}};
    const size_t P = m->h_cat.size();
    for (size_t j = 0; j < P; ++j) {
        const int cat = m->h_cat[j];
        if (cat != 1) {
            const double sl = m->h_static[j];
            const double tcut = cat == 0 ? thr + m->h_m0[j] : (thr + m->h_m0[j]) + m->h_m1[j];
            if (sl - 1e-9 * (1.0 + fabs(sl)) >= tcut) continue;
        }
        c.n[m->h_cls[j]] += 1;
    }
    return c;
}
static inline unsigned long long sub_queue_cap(const nbk_model* m, const PairCounts& pc, unsigned long long nblk) {
    unsigned long long cap = WAVE;
    for (int c = 0; c < 4; ++c) {
        if (pc.n[c] == 0) continue;
        const unsigned long long g = (unsigned long long)m->d.cls_groups[c];
        const unsigned long long v = ((nblk + g - 1) / g) * WAVE * (unsigned long long)pc.n[c];
        if (v > cap) cap = v;
    }
    return cap;
}

// Queue sizing.  Robots that fit the LDS-parked layout (the queue-less kernel can re-decide a block): one tile of up to TILE_MAX
// configurations, every sub-queue as large as the worst case needs but at most its share of WS_MAX_BYTES -- blocks whose items do
// not fit are re-decided by k_validity_redo.  Larger robots: tiles small enough for the worst case (every pair of every
// configuration), as nothing can catch an overflow for them.
static inline int64_t tile_configs(const nbk_model* m, const PairCounts& pc, int64_t B) {
    const int64_t Bp = ((B + WAVE - 1) / WAVE) * WAVE;
    if (m->parked_ok) return Bp < TILE_MAX ? Bp : TILE_MAX;
    double per_cfg = 1.0;
    for (int c = 0; c < 4; ++c) if (pc.n[c] > 0) { const double v = (double)NSUB * pc.n[c] / m->d.cls_groups[c]; if (v > per_cfg) per_cfg = v; }
    const int64_t P = (int64_t)per_cfg + 1;
    const size_t ws_max = size_t(8) << 30;
    int64_t t = (int64_t)((ws_max - ws_header(m)) / (8 * (size_t)P)) - (int64_t)NSUB * WAVE;
    t = (t / WAVE) * WAVE;
    if (t < 2 * (int64_t)NSUB * WAVE) t = 2 * (int64_t)NSUB * WAVE;
    if (t > TILE_MAX) t = TILE_MAX;
    return Bp < t ? Bp : t;
}
static inline unsigned long long tile_queue_cap(const nbk_model* m, const PairCounts& pc, unsigned long long nblk) {
    const unsigned long long worst = sub_queue_cap(m, pc, nblk);
    if (!m->parked_ok) return worst;
    size_t bytes = g_opt.queue_budget > 0 ? (size_t)g_opt.queue_budget : WS_MAX_BYTES;
    if (bytes > 2 * ws_header(m)) bytes -= ws_header(m);           // the whole workspace, tables included, stays within the budget
    unsigned long long budget = (unsigned long long)(bytes / (8 * (size_t)NSUB));
    if (budget < (unsigned long long)WAVE) budget = WAVE;
    return worst < budget ? worst : budget;
}

static inline size_t broad_reg_lds(const nbk_model* m, int S) {
    const size_t qrows = ((size_t)WAVE * m->d.n_q * 8 >= (size_t)BQ_CAP * 4) ? (size_t)m->d.n_q : ((size_t)BQ_CAP * 4 + WAVE * 8 - 1) / (WAVE * 8);
    const size_t W = (size_t)m->d.n_wshapes;
    return sizeof(double) * (WAVE * (qrows + 12 * (size_t)m->d.frame_slots) + (size_t)S * S + 2 * W * S) + sizeof(int) * ((size_t)S * S + W * S) + 16;
}

// the scratch set of (descriptor, stream); created on the stream's first call.  nullptr: too many streams (use the _ws variant)
static StreamWs* stream_ws(nbk_model* mm, hipStream_t st) {
    std::lock_guard<std::mutex> lock(mm->mu);
    for (StreamWs* w : mm->wss) if (w->stream == st) return w;
    if (mm->wss.size() >= 64) return nullptr;
    StreamWs* w = new StreamWs();
    w->stream = st; w->ws = nullptr; w->ws_bytes = 0; w->ready = false; w->thr = 0.0; w->epoch = 0; w->captured = false;
    w->ews = nullptr; w->ews_bytes = 0; w->ecap_edges = 0; w->ecap_samples = 0; w->stats = nullptr; w->stats_dev = nullptr;
    w->aux_stream = nullptr; w->ev_fork = nullptr; w->ev_join = nullptr; w->aux = nullptr;
    mm->wss.push_back(w);
    return w;
}

// (re)allocate a scratch buffer of this stream's set.  Earlier work of THIS stream may still read the old buffer: wait for it
// (other streams never touch it).  Never called while capturing.
static int32_t grow_scratch(hipStream_t st, void*& buf, size_t& have, size_t need, const char* what) {
    if (have >= need) return NBK_OK;
    if (buf) { NBK_HIP(hipStreamSynchronize(st)); (void)hipFree(buf); buf = nullptr; have = 0; }
    hipError_t e = hipMalloc(&buf, need);
    if (e != hipSuccess) { hip_fail(e, what); return NBK_ERR_ALLOC; }
    have = need;
    return NBK_OK;
}

// broadphase + narrowphase over B configurations (plain q rows, or the samples described by `es`), tiled so that the worst-case
// queue fits the workspace.  `iw`: the workspace is this stream's own set and keeps state between calls (tables, counter epoch);
// nullptr: caller-owned workspace, or a call being captured into a graph -- self-contained: every call prepares its tables
// and clears its counters itself.
// tile size of pipelined batches (NBK_PIPE_TILE): whole 64-configuration blocks (tiles on the two streams must not share a mask word
// or a block), at least one block per sub-queue; anything else is rounded / clamped here, so no value of the switch changes a result
static inline int64_t pipe_tile_configs() {
    int64_t t = g_opt.pipe_tile > 0 ? (int64_t)g_opt.pipe_tile : (int64_t(1) << 20);
    t &= ~int64_t(WAVE - 1);
    const int64_t lo = (int64_t)NSUB * WAVE;
    return t < lo ? lo : t;
}
#define PIPE_TILE pipe_tile_configs()
static inline bool pipelined(const nbk_model* m, int64_t B) { return g_opt.pipeline_tiles != 0 && m->parked_ok && B >= 2 * PIPE_TILE; }
static inline int64_t call_tile(const nbk_model* m, const PairCounts& pc, int64_t B, bool pipe) {
    const int64_t t = tile_configs(m, pc, B);
    return pipe && t > PIPE_TILE ? PIPE_TILE : t;
}

// which GJK walks can a call at this threshold need?  tc = (thr + mA) + mB per pair that can reach GJK: all zero and no hull -> the
// boolean walk only (0: k_narrow_bool); none negative -> the (inflated) walk + the distance iteration for hulls and undecided walks
// (1: k_narrow_pos); all negative -> the distance predicate only (2: k_narrow_pred); else the build with everything (3: k_narrow)
static int narrow_variant(const nbk_model* m, double threshold) {
    bool any_zero = false, any_positive = false, any_negative = false;
    for (size_t i = 0; i + 1 < m->gjk_margins.size(); i += 2) {
        const double tc = (threshold + m->gjk_margins[i]) + m->gjk_margins[i + 1];
        if (tc == 0.0) any_zero = true;
        else if (tc > 0.0) any_positive = true;
        else any_negative = true;                      // (a NaN threshold counts as negative: the distance predicate)
    }
    if (!any_positive && !any_negative && !m->gjk_any_hull) return 0;
    if (!any_negative) return 1;
    if (!any_zero && !any_positive) return 2;
    return 3;
}
// diagnostic (not part of include/nbk.h): the narrowphase build nbk_validity_batch picks for this descriptor at this threshold
extern "C" int32_t nbk_debug_narrow_variant(const nbk_model* m, double threshold) { return m == nullptr ? NBK_ERR_INVALID : narrow_variant(m, threshold); }

// `pipe` (the library's own scratch only): odd tiles run on iw0->aux_stream with the scratch set iw0->aux
static int32_t launch_two_kernel_impl(const nbk_model* m, const PairCounts& pc, EdgeSrc es, const double* q, int64_t B, double threshold, uint64_t* mask_bits,
                                      uint8_t* mask_bytes, void* workspace0, hipStream_t st0, StreamWs* iw0, bool pipe) {
    const int64_t tile = call_tile(m, pc, B, pipe);
    if (pipe) {
        NBK_HIP(hipEventRecord(iw0->ev_fork, st0));                          // the odd tiles' inputs are whatever the caller's stream has produced
        NBK_HIP(hipStreamWaitEvent(iw0->aux_stream, iw0->ev_fork, 0));
    }
    // which GJK walks can this call need?  tc = (thr + mA) + mB per pair: all zero and no hull -> boolean walk only, none negative ->
    // the (inflated) walk + the distance iteration for hulls and undecided walks, all negative -> distance predicate only, else
    // the build with everything
    const int narrow_build = narrow_variant(m, threshold);
    const int S = m->d.n_rshapes;
    const bool use_reg = S <= 16 && (!g_opt.no_reg_broad || !m->lds_broad_ok);
    const bool f32 = !g_opt.f64_broad || broad_reg_lds(m, S <= 8 ? 8 : (S <= 12 ? 12 : 16)) > 160 * 1024;   // the float64 form keeps its tables in LDS
    int tile_no = 0;
    for (int64_t b0 = 0; b0 < B; b0 += tile, ++tile_no) {
        const bool odd = pipe && (tile_no & 1);
        StreamWs* iw = odd ? iw0->aux : iw0;
        hipStream_t st = odd ? iw0->aux_stream : st0;
        void* workspace = odd ? iw0->aux->ws : workspace0;
        const bool internal = iw != nullptr;
        unsigned long long* count_set0 = static_cast<unsigned long long*>(workspace);
        unsigned long long* count = count_set0;
        unsigned long long* items = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + ws_header(m));
        const int64_t nb = (B - b0) < tile ? (B - b0) : tile;
        const unsigned nblk = blocks_for(nb);
        // a sub-queue takes one kind class of the blocks of one group (every 64th block): worst case all pairs of that class
        const unsigned long long cap_sub = tile_queue_cap(m, pc, nblk);
        EdgeSrc es_tile = es;
        if (es.map != nullptr) { es_tile.map = es.map + b0; es_tile.b0 = b0; }
        es_tile.ovf = m->parked_ok ? reinterpret_cast<unsigned char*>(workspace) + ws_tables(m) : nullptr;
        unsigned long long* flag_words = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + ws_tables(m));
        const int n_flag_words = m->parked_ok ? (int)((nblk + 7) / 8) : 0;
        // tiles start on a multiple of 64 configurations, so mask words never straddle tiles
        const double* qt = q ? q + b0 * m->n_q : nullptr;
        uint64_t* mb = mask_bits ? mask_bits + b0 / 64 : nullptr;
        uint8_t* my = mask_bytes ? mask_bytes + b0 : nullptr;
        float* ftab = reinterpret_cast<float*>(static_cast<char*>(workspace) + WS_COUNTERS);
        // LDS of the float32 kernel: q slab (later the item queue) + saved frames
        const size_t qrows_f = (size_t)f32_qrows(m->d.n_q, S <= 8 ? 8 : (S <= 12 ? 12 : 16));
        const size_t lds_f = sizeof(double) * WAVE * qrows_f + sizeof(float) * WAVE * 12 * (size_t)m->d.frame_slots + 16 + sizeof(float) * WAVE * NBK_ZSLOTS;   // (+ the z coordinates of the slots k_broad_f32 keeps in LDS)
        unsigned long long* count_next = nullptr;
        if (use_reg && f32) {
            if (internal && iw->ready && !iw->captured && iw->thr == threshold) {
                // tables are in place and the previous call's narrowphase cleared this call's counter set: no launch
                count = count_set0 + (size_t)(iw->epoch & 1u) * NSUB * CNT_STRIDE;
                count_next = count_set0 + (size_t)((iw->epoch + 1u) & 1u) * NSUB * CNT_STRIDE;
            } else {
                count = count_set0;
                hipLaunchKernelGGL(k_prepare_f32, dim3(1), dim3(NSUB), 0, st, m->d, threshold, count_set0, internal ? 2 : 1, ftab, flag_words, n_flag_words);   // clears the counters too
                if (internal) { iw->ready = true; iw->thr = threshold; iw->epoch = 0; count_next = count_set0 + (size_t)NSUB * CNT_STRIDE; }
            }
            if (internal) iw->epoch += 1u;
        } else {
            count = count_set0;
            if (internal) iw->ready = false;
            hipLaunchKernelGGL(k_zero_counters, dim3(1), dim3(NSUB), 0, st, count, flag_words, n_flag_words);
        }
#define NBK_LAUNCH_BF32(S_, WH_) hipLaunchKernelGGL((k_broad_f32<S_, WH_>), dim3(nblk), dim3(WAVE), lds_f, st, m->d, es_tile, qt, nb, threshold, mb, my, count, items, cap_sub, ftab)
        if (use_reg && f32 && S <= 8) { if (m->world_hulls) NBK_LAUNCH_BF32(8, true); else NBK_LAUNCH_BF32(8, false); }
        else if (use_reg && f32 && S <= 12) { if (m->world_hulls) NBK_LAUNCH_BF32(12, true); else NBK_LAUNCH_BF32(12, false); }
        else if (use_reg && f32) { if (m->world_hulls) NBK_LAUNCH_BF32(16, true); else NBK_LAUNCH_BF32(16, false); }
#undef NBK_LAUNCH_BF32
        else if (use_reg && S <= 8)
            hipLaunchKernelGGL(k_broad_reg<8>, dim3(nblk), dim3(WAVE), broad_reg_lds(m, 8), st, m->d, es_tile, qt, nb, threshold, mb, my, count, items, cap_sub);
        else if (use_reg && S <= 12)
            hipLaunchKernelGGL(k_broad_reg<12>, dim3(nblk), dim3(WAVE), broad_reg_lds(m, 12), st, m->d, es_tile, qt, nb, threshold, mb, my, count, items, cap_sub);
        else if (use_reg)
            hipLaunchKernelGGL(k_broad_reg<16>, dim3(nblk), dim3(WAVE), broad_reg_lds(m, 16), st, m->d, es_tile, qt, nb, threshold, mb, my, count, items, cap_sub);
        else
            hipLaunchKernelGGL(k_broad, dim3(nblk), dim3(WAVE), broad_lds(m), st, m->d, es_tile, qt, nb, threshold, mb, my, count, items, cap_sub);
        NBK_HIP(hipGetLastError());
        const size_t nlds = sizeof(double) * NARROW_T * (size_t)m->n_q + narrow_hull_lds(m->d.hull_blob_n);
        // workgroups per sub-queue: one 64-item chunk each at a few survivors per configuration; more chunks are strided over
        unsigned parts = 4u * nblk / NSUB;
        { const unsigned pmax = g_opt.narrow_parts_max > 0 ? (unsigned)g_opt.narrow_parts_max : 16u; parts = parts < 4u ? 4u : parts; parts = parts > pmax ? pmax : parts; }
        if (nblk <= 4u) parts = 1u;                      // a handful of configurations (the scalar calls): 256 workgroups are plenty
        if (narrow_build == 0)
            hipLaunchKernelGGL(k_narrow_bool, dim3(NSUB * parts), dim3(NARROW_T), nlds, st, m->d, es_tile, qt, threshold, items, count, cap_sub, mb, my, count_next);
        else if (narrow_build == 1)
            hipLaunchKernelGGL(k_narrow_pos, dim3(NSUB * parts), dim3(NARROW_T), nlds, st, m->d, es_tile, qt, threshold, items, count, cap_sub, mb, my, count_next);
        else if (narrow_build == 2)
            hipLaunchKernelGGL(k_narrow_pred, dim3(NSUB * parts), dim3(NARROW_T), nlds, st, m->d, es_tile, qt, threshold, items, count, cap_sub, mb, my, count_next);
        else
            hipLaunchKernelGGL(k_narrow, dim3(NSUB * parts), dim3(NARROW_T), nlds, st, m->d, es_tile, qt, threshold, items, count, cap_sub, mb, my, count_next);
        NBK_HIP(hipGetLastError());
        // blocks whose items overflowed their sub-queue (possible only when the budget, not the worst case, sized the queue)
        if (m->parked_ok && cap_sub < sub_queue_cap(m, pc, nblk)) {
            hipLaunchKernelGGL(k_validity_redo, dim3(nblk), dim3(WAVE), collide_lds(m), st, m->d, es_tile, qt, nb, threshold, mb, my);
            NBK_HIP(hipGetLastError());
        }
    }
    if (pipe) {
        NBK_HIP(hipEventRecord(iw0->ev_join, iw0->aux_stream));              // the caller's stream continues when the odd tiles are done too
        NBK_HIP(hipStreamWaitEvent(st0, iw0->ev_join, 0));
    }
    return NBK_OK;
}

static int32_t launch_two_kernel(const nbk_model* m, const PairCounts& pc, EdgeSrc es, const double* q, int64_t B, double threshold, uint64_t* mask_bits,
                                 uint8_t* mask_bytes, void* workspace, hipStream_t st, StreamWs* iw = nullptr, bool pipe = false) {
    const int32_t rc = launch_two_kernel_impl(m, pc, es, q, B, threshold, mask_bits, mask_bytes, workspace, st, iw, pipe);
    if (rc != NBK_OK && iw != nullptr) { iw->ready = false; if (iw->aux) iw->aux->ready = false; }      // whatever state the queues are in: start over
    return rc;
}

// the second stream, its events and its scratch set of a pipelined call (created on first use; never while capturing)
static int32_t pipe_setup(nbk_model* mm, const nbk_model* m, const PairCounts& pc, StreamWs* w, int64_t B, hipStream_t st) {
    if (w->aux_stream == nullptr) {
        NBK_HIP(hipStreamCreateWithFlags(&w->aux_stream, hipStreamNonBlocking));
        NBK_HIP(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming));
        NBK_HIP(hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming));
        w->aux = stream_ws(mm, w->aux_stream);
        if (w->aux == nullptr) return NBK_ERR_ALLOC;
    }
    const int64_t nblk = (call_tile(m, pc, B, true) + WAVE - 1) / WAVE;
    const size_t need = ws_header(m) + 8 * (size_t)NSUB * (size_t)tile_queue_cap(m, pc, (unsigned long long)nblk);
    if (w->aux->ws_bytes < need) {
        w->aux->ready = false;
        const int32_t rc = grow_scratch(w->aux_stream, w->aux->ws, w->aux->ws_bytes, need, "hipMalloc(workspace, second stream)");
        if (rc != NBK_OK) return rc;
        NBK_HIP(hipMemsetAsync(static_cast<char*>(w->aux->ws) + ws_tables(m), 0, WS_FLAGS, w->aux_stream));
    }
    (void)st;
    return NBK_OK;
}

static int64_t two_kernel_workspace_bytes(const nbk_model* m, const PairCounts& pc, int64_t B, bool pipe = false) {
    const int64_t nblk = (call_tile(m, pc, B, pipe) + WAVE - 1) / WAVE;
    return (int64_t)ws_header(m) + 8 * (int64_t)NSUB * (int64_t)tile_queue_cap(m, pc, (unsigned long long)nblk);
}

int64_t nbk_validity_workspace_bytes(const nbk_model* m, int64_t B) {
    if (m == nullptr || B < 0) return NBK_ERR_INVALID;
    if ((B < g_opt.two_kernel_min_b && m->parked_ok) || m->n_pairs == 0 || B == 0) return 0;
    // NSUB sub-queues, each sized for the blocks that map to it (rounded up)
    return two_kernel_workspace_bytes(m, all_pairs(m), B);        // the caller's workspace serves every threshold
}

static const EdgeSrc NO_EDGES = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr};

int32_t nbk_validity_batch_ws(const nbk_model* m, const double* q, int64_t B, double threshold, uint64_t* mask_bits,
                              uint8_t* mask_bytes, void* workspace, int64_t workspace_bytes, void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && q == nullptr) || (mask_bits == nullptr && mask_bytes == nullptr)) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (B == 0) return NBK_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t need = nbk_validity_workspace_bytes(m, B);
    if (workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 63u) != 0) {
        snprintf(g_err, sizeof(g_err), "workspace must be 64-byte aligned");
        return NBK_ERR_INVALID;
    }
    if (need == 0 || workspace == nullptr || workspace_bytes < need) {
        if (need != 0 && workspace != nullptr) return NBK_ERR_INVALID;       // a workspace was given but is too small
        if (!m->parked_ok) return need != 0 ? NBK_ERR_INVALID : NBK_ERR_UNSUPPORTED;   // this robot needs the workspace path
        hipLaunchKernelGGL(k_validity, dim3(blocks_for(B)), dim3(WAVE), collide_lds(m), st, m->d, q, B, threshold, mask_bits, mask_bytes);
        NBK_HIP(hipGetLastError());
        return NBK_OK;
    }
    return launch_two_kernel(m, all_pairs(m), NO_EDGES, q, B, threshold, mask_bits, mask_bytes, workspace, st);
}

int32_t nbk_validity_batch(const nbk_model* m, const double* q, int64_t B, double threshold, uint64_t* mask_bits,
                           uint8_t* mask_bytes, void* stream) {
    if (m == nullptr) return NBK_ERR_INVALID;
    if (nbk_validity_workspace_bytes(m, B) <= 0) return nbk_validity_batch_ws(m, q, B, threshold, mask_bits, mask_bytes, nullptr, 0, stream);
    if (B < 0 || q == nullptr || (mask_bits == nullptr && mask_bytes == nullptr)) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    // the library's own scratch is sized for the pairs that are within reach at THIS threshold (obstacle-rich scenes: most world
    // shapes are out of the arm's reach for good)
    const PairCounts pc = reachable_pairs(m, threshold);
    hipStream_t st = (hipStream_t)stream;
    const bool capturing = stream_capturing(st);
    const bool pipe = pipelined(m, B) && !capturing;
    const int64_t need = two_kernel_workspace_bytes(m, pc, B, pipe);
    StreamWs* w = stream_ws(const_cast<nbk_model*>(m), st);
    if (w == nullptr) { snprintf(g_err, sizeof(g_err), "more than 64 streams use this descriptor's internal workspaces: pass your own (nbk_validity_batch_ws)"); return NBK_ERR_ALLOC; }
    std::lock_guard<std::mutex> lock(w->mu);
    if (w->ws_bytes < (size_t)need) {
        if (capturing) {
            snprintf(g_err, sizeof(g_err), "graph capture: this stream's internal workspace is not allocated yet -- run the call once "
                     "outside the capture, or pass a workspace (nbk_validity_batch_ws)");
            return NBK_ERR_UNSUPPORTED;
        }
        w->ready = false;
        const int32_t rc = grow_scratch(st, w->ws, w->ws_bytes, (size_t)need, "hipMalloc(workspace)");
        if (rc != NBK_OK) return rc;
        NBK_HIP(hipMemsetAsync(static_cast<char*>(w->ws) + ws_tables(m), 0, WS_FLAGS, st));      // no overflow marks yet
    }
    // a captured call must be self-contained (it is replayed out of order with the host-side state): prepare + clear inside
    // the graph, and the next direct call starts from scratch as well
    if (capturing) { w->ready = false; w->captured = true; }
    if (pipe) { const int32_t rc = pipe_setup(const_cast<nbk_model*>(m), m, pc, w, B, st); if (rc != NBK_OK) return rc; }
    return launch_two_kernel(m, pc, NO_EDGES, q, B, threshold, mask_bits, mask_bytes, w->ws, st, capturing ? nullptr : w, pipe);
}

// workgroups that share the pair list of one block of configurations in the per-pair distance kernels: enough to put ~8 waves on
// every SIMD pair of the chip when the batch alone would not (at most one group per pair)
static inline unsigned pair_groups(const nbk_model* m, int64_t B) {
    const long long nblk = (B + WAVE - 1) / WAVE;
    long long g = (4096 + nblk - 1) / nblk;
    if (g > m->n_pairs) g = m->n_pairs;
    if (g > 64) g = 64;
    return (unsigned)(g < 1 ? 1 : g);
}

// the two-wave workgroups of k_distances<1..3>: half as many groups for the same number of waves
static inline unsigned pair_groups2(const nbk_model* m, int64_t B) { const unsigned g = pair_groups(m, B); return g > 1 ? (g + 1) / 2 : 1; }

int32_t nbk_closest_batch(const nbk_model* m, const double* q, int64_t B, double* min_dist, int32_t* argmin, void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || min_dist == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (!m->parked_ok) return NBK_ERR_UNSUPPORTED;
    if (B == 0) return NBK_OK;
    // branch-and-bound needs its result / best / argmin / queue / EPA-list arrays next to the parked cores; a robot that leaves no room
    // for them gets every pair evaluated (same result)
    const size_t closest_lds = collide_lds(m) - VALIDITY_LDS_EXTRA + sizeof(double) * CQ_CAP + 8 * WAVE + 4 * WAVE + 4 * CQ_CAP + 2 * CQ_CAP;
    if (g_opt.closest_brute || closest_lds > 160 * 1024)
        hipLaunchKernelGGL(k_distances<0>, dim3(blocks_for(B)), dim3(WAVE), collide_lds(m), (hipStream_t)stream, m->d, q, B, min_dist,
                           argmin, (double*)nullptr);
    else
        hipLaunchKernelGGL(k_closest, dim3(blocks_for(B)), dim3(WAVE), closest_lds, (hipStream_t)stream, m->d, q, B, min_dist, argmin);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

int32_t nbk_pair_distances_batch(const nbk_model* m, const double* q, int64_t B, double* dist, double* witness, void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || dist == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (!m->parked_ok || collide_lds(m) + 8 * EPAQ_DOUBLES > 160 * 1024) return NBK_ERR_UNSUPPORTED;
    if (B == 0 || m->n_pairs == 0) return NBK_OK;
    if (witness != nullptr)
        hipLaunchKernelGGL(k_distances<2>, dim3(blocks_for(B), pair_groups2(m, B)), dim3(2 * WAVE), collide_lds(m) + 8 * EPAQ_DOUBLES, (hipStream_t)stream, m->d, q, B, dist,
                           (int32_t*)nullptr, witness);
    else
        hipLaunchKernelGGL(k_distances<1>, dim3(blocks_for(B), pair_groups2(m, B)), dim3(2 * WAVE), collide_lds(m) + 8 * EPAQ_DOUBLES, (hipStream_t)stream, m->d, q, B, dist,
                           (int32_t*)nullptr, (double*)nullptr);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

int32_t nbk_proximity_jacobian_batch(const nbk_model* m, const double* q, int64_t B, double* dist, double* witness, double* jrows,
                                     void* stream) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || dist == nullptr || witness == nullptr || jrows == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (!m->parked_ok) return NBK_ERR_UNSUPPORTED;
    if (B == 0 || m->n_pairs == 0) return NBK_OK;
    const size_t lds = collide_lds(m) + sizeof(double) * WAVE * 6 * (size_t)m->n_joints + 8 * EPAQ_DOUBLES;
    if (lds > 160 * 1024) return NBK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_distances<3>, dim3(blocks_for(B), pair_groups2(m, B)), dim3(2 * WAVE), lds, (hipStream_t)stream, m->d, q, B, dist, (int32_t*)nullptr,
                       witness, jrows);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

// ---- batched DiscreteConnector: fully asynchronous -----------------------------------------------------------------------------
// The sample count of an edge batch is only known on the device (the lengths come from device arrays), so nothing here waits
// for it: the stream's edge scratch has a CAPACITY in samples, every launch covers the capacity and blocks beyond the true
// count exit at once (EdgeSrc::total).  Capacity = E x (ceil(max_distance / resolution) + 2) samples at least -- exact for
// steer, and for connect whenever the caller's edges respect the connector's max_distance (planners do) -- and at least 1.25 x
// the count the previous call reported through pinned memory (`stats`, read here without waiting).  Edges that do not fit
// anyway are marked and walked by one wave each (k_edges_overflow): always correct, only slower.
static inline unsigned long long edge_capacity(int64_t E, double resolution, double max_distance) {
    double per = ceil(max_distance / resolution) + 2.0;
    if (!(per < 4096.0)) per = 4096.0;                    // an unbounded max_distance: start from 4096 samples per edge
    double c = (double)E * per;
    if (c < 4096.0) c = 4096.0;
    if (c > 4.0e9) c = 4.0e9;
    return ((unsigned long long)c + 63ull) & ~63ull;
}

int32_t nbk_edge_validity_batch(const nbk_model* m, const double* starts, const double* goals, const double* dist, int64_t E,
                                double resolution, double max_distance, int32_t mode, double threshold, uint8_t* valid,
                                double* end, int32_t* n_samples, void* stream) {
    if (m == nullptr || E < 0 || (E > 0 && (starts == nullptr || goals == nullptr || valid == nullptr))) return NBK_ERR_INVALID;
    if (!(resolution > 0.0) || !(max_distance > 0.0) || (mode != NBK_CONNECT && mode != NBK_STEER)) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (E == 0) return NBK_OK;
    if (E > 0x7fffffffLL) return NBK_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if ((E < g_opt.edge_batch_min_e && m->parked_ok) || m->n_pairs == 0) {
        // one wave per edge, early exit, one launch
        hipLaunchKernelGGL(k_edges, dim3((unsigned)E), dim3(WAVE), collide_lds(m), st, m->d, starts, goals, dist, E,
                           resolution, max_distance, mode, threshold, valid, end, n_samples, (const uint8_t*)nullptr, (unsigned long long*)nullptr);
        NBK_HIP(hipGetLastError());
        return NBK_OK;
    }
    StreamWs* w = stream_ws(const_cast<nbk_model*>(m), st);
    if (w == nullptr) { snprintf(g_err, sizeof(g_err), "more than 64 streams use this descriptor's internal workspaces"); return NBK_ERR_ALLOC; }
    std::lock_guard<std::mutex> lock(w->mu);
    const bool capturing = stream_capturing(st);
    // capacity: the static bound, and what earlier calls on this stream turned out to need (pinned memory, not waited for)
    unsigned long long cap = edge_capacity(E, resolution, max_distance);
    if (w->stats != nullptr && __atomic_load_n(&w->stats[1], __ATOMIC_RELAXED) != 0ull) {
        // the last finished call had edges that did not fit: leave headroom over what it needed
        const unsigned long long seen = __atomic_load_n(&w->stats[0], __ATOMIC_RELAXED);
        const unsigned long long want = seen + seen / 4;
        if (want > cap && want < 4000000000ull) cap = (want + 63ull) & ~63ull;
    }
    if (!m->parked_ok) {
        // robots whose primitives do not fit the one-wave-per-edge kernel cannot serve overflowing edges: such descriptors
        // keep a synchronous sizing step (one read-back) and cannot be captured
        if (capturing) { snprintf(g_err, sizeof(g_err), "graph capture of edge batches needs a robot that fits the LDS-parked layout"); return NBK_ERR_UNSUPPORTED; }
    }
    const PairCounts pc = reachable_pairs(m, threshold);
    double* plan = nullptr;
    unsigned long long *cnt = nullptr, *offs = nullptr, *map = nullptr;
    uint8_t* ovf = nullptr;
    uint64_t* words = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool fits = w->ecap_edges >= E && w->ecap_samples >= cap && w->stats != nullptr &&
                          w->ws_bytes >= (size_t)two_kernel_workspace_bytes(m, pc, (int64_t)w->ecap_samples);
        if (!fits) {
            if (capturing) {
                snprintf(g_err, sizeof(g_err), "graph capture: this stream's edge scratch is not allocated for %lld edges yet -- run the "
                         "call once outside the capture", (long long)E);
                return NBK_ERR_UNSUPPORTED;
            }
            if (w->stats == nullptr) {
                NBK_HIP(hipHostMalloc((void**)&w->stats, 4 * sizeof(unsigned long long), hipHostMallocMapped));
                memset(w->stats, 0, 4 * sizeof(unsigned long long));
                NBK_HIP(hipHostGetDevicePointer((void**)&w->stats_dev, w->stats, 0));
            }
            const long long ne = w->ecap_edges > E ? w->ecap_edges : E;
            const unsigned long long nc = w->ecap_samples > cap ? w->ecap_samples : cap;
            const size_t head_n = ((size_t)ne * 3 * 8 + (size_t)(ne + 1) * 8 * 2 + (size_t)ne + 4095) & ~size_t(4095);   // plan | cnt | offs | overflow flags
            const size_t map_n = ((size_t)nc * 8 + 4095) & ~size_t(4095);
            const size_t words_n = (((size_t)nc + 63) / 64 * 8 + 4095) & ~size_t(4095);
            int32_t rc = grow_scratch(st, w->ews, w->ews_bytes, head_n + map_n + words_n, "hipMalloc(edge scratch)");
            if (rc != NBK_OK) return rc;
            w->ecap_edges = ne; w->ecap_samples = nc;
            const size_t need = (size_t)two_kernel_workspace_bytes(m, pc, (int64_t)nc);
            if (w->ws_bytes < need) {
                w->ready = false;
                rc = grow_scratch(st, w->ws, w->ws_bytes, need, "hipMalloc(workspace)");
                if (rc != NBK_OK) return rc;
                NBK_HIP(hipMemsetAsync(static_cast<char*>(w->ws) + ws_tables(m), 0, WS_FLAGS, st));
            }
        }
        cap = w->ecap_samples;                                 // use all of what is there
        const size_t head_c = ((size_t)w->ecap_edges * 3 * 8 + (size_t)(w->ecap_edges + 1) * 8 * 2 + (size_t)w->ecap_edges + 4095) & ~size_t(4095);
        const size_t map_c = ((size_t)cap * 8 + 4095) & ~size_t(4095);
        plan = static_cast<double*>(w->ews);
        cnt = reinterpret_cast<unsigned long long*>(plan + 3 * w->ecap_edges);
        offs = cnt + (w->ecap_edges + 1);
        ovf = reinterpret_cast<uint8_t*>(offs + (w->ecap_edges + 1));
        map = reinterpret_cast<unsigned long long*>(static_cast<char*>(w->ews) + head_c);
        words = reinterpret_cast<uint64_t*>(static_cast<char*>(w->ews) + head_c + map_c);
        hipLaunchKernelGGL(k_edge_plan, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, m->n_q, starts, goals, dist, E, resolution,
                           max_distance, mode, plan, cnt, end, n_samples);
        NBK_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, cnt, E, offs);
        NBK_HIP(hipGetLastError());
        if (m->parked_ok) break;
        // (see above) no overflow kernel for this robot: size exactly, with one read-back
        unsigned long long total = 0;
        NBK_HIP(hipMemcpyAsync(&total, offs + E, 8, hipMemcpyDeviceToHost, st));
        NBK_HIP(hipStreamSynchronize(st));
        if (total <= cap) break;
        if (attempt == 1 || total >= 4000000000ull) { snprintf(g_err, sizeof(g_err), "edge batch of %llu samples is too large", total); return NBK_ERR_UNSUPPORTED; }
        cap = (total + 63ull) & ~63ull;
    }
    hipLaunchKernelGGL(k_edge_expand, dim3((unsigned)E), dim3(WAVE), 0, st, offs, E, map, cap, ovf);
    NBK_HIP(hipGetLastError());
    EdgeSrc es{starts, goals, plan, map, offs + E, 0, nullptr};
    if (capturing) { w->ready = false; w->captured = true; }
    const bool pipe = pipelined(m, (int64_t)cap) && !capturing;
    if (pipe) { const int32_t rp = pipe_setup(const_cast<nbk_model*>(m), m, pc, w, (int64_t)cap, st); if (rp != NBK_OK) return rp; }
    const int32_t rc = launch_two_kernel(m, pc, es, nullptr, (int64_t)cap, threshold, words, nullptr, w->ws, st, capturing ? nullptr : w, pipe);
    if (rc != NBK_OK) return rc;
    hipLaunchKernelGGL(k_edge_reduce, dim3((unsigned)E), dim3(WAVE), 0, st, offs, E, words, ovf, valid, w->stats_dev);
    NBK_HIP(hipGetLastError());
    if (m->parked_ok) {
        hipLaunchKernelGGL(k_edges, dim3((unsigned)E), dim3(WAVE), collide_lds(m), st, m->d, starts, goals, dist, E,
                           resolution, max_distance, mode, threshold, valid, (double*)nullptr, (int32_t*)nullptr, (const uint8_t*)ovf, w->stats_dev);
        NBK_HIP(hipGetLastError());
    }
    return NBK_OK;
}

// ---- scalar calls from host memory (Arm.in_collision(q), DiscreteConnector.connect(a, b) on one configuration / one edge) ----
// The reference's planners call these once per sample / per edge (numbotics/planning/sampling_based/prm.py:40,
// connectors.py:57-100), so what counts is latency: inputs and results travel through pinned host memory that the kernels
// read and write directly (no staging copies), on a private stream with its own scratch set; one wait at the end.
static int32_t scalar_setup(nbk_model* mm) {
    if (mm->scalar_q != nullptr) return NBK_OK;
    const size_t nd = 4 * (size_t)mm->n_q + 8;
    NBK_HIP(hipHostMalloc((void**)&mm->scalar_q, nd * sizeof(double), hipHostMallocMapped));
    NBK_HIP(hipHostMalloc((void**)&mm->scalar_out, 8 * sizeof(unsigned long long), hipHostMallocMapped));
    NBK_HIP(hipStreamCreateWithFlags(&mm->scalar_stream, hipStreamNonBlocking));
    NBK_HIP(hipHostGetDevicePointer((void**)&mm->scalar_q_dev, mm->scalar_q, 0));
    NBK_HIP(hipHostGetDevicePointer((void**)&mm->scalar_out_dev, mm->scalar_out, 0));
    return NBK_OK;
}

int32_t nbk_validity_scalar_host(const nbk_model* m, const double* q, double threshold, int32_t* in_collision) {
    if (m == nullptr || q == nullptr || in_collision == nullptr) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    nbk_model* mm = const_cast<nbk_model*>(m);
    std::lock_guard<std::mutex> lock(mm->scalar_mu);
    { const int32_t rc = scalar_setup(mm); if (rc != NBK_OK) return rc; }
    memcpy(mm->scalar_q, q, sizeof(double) * (size_t)m->n_q);
    double* dq = mm->scalar_q_dev;
    uint64_t* dout = reinterpret_cast<uint64_t*>(mm->scalar_out_dev);
    mm->scalar_out[0] = 0ull;
    const int32_t rc = nbk_validity_batch(m, dq, 1, threshold, dout, nullptr, mm->scalar_stream);
    if (rc != NBK_OK) return rc;
    NBK_HIP(hipStreamSynchronize(mm->scalar_stream));
    *in_collision = (int32_t)(mm->scalar_out[0] & 1ull);
    return NBK_OK;
}

int32_t nbk_edge_validity_scalar_host(const nbk_model* m, const double* start, const double* goal, double dist, double resolution,
                                      double max_distance, int32_t mode, double threshold, int32_t* valid, double* end,
                                      int32_t* n_samples) {
    if (m == nullptr || start == nullptr || goal == nullptr || valid == nullptr) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    nbk_model* mm = const_cast<nbk_model*>(m);
    std::lock_guard<std::mutex> lock(mm->scalar_mu);
    { const int32_t rc = scalar_setup(mm); if (rc != NBK_OK) return rc; }
    const size_t nq = (size_t)m->n_q;
    double* h = mm->scalar_q;                           // start | goal | end | dist
    memcpy(h, start, sizeof(double) * nq);
    memcpy(h + nq, goal, sizeof(double) * nq);
    h[3 * nq] = dist;
    double* d = mm->scalar_q_dev;
    unsigned long long* dout = mm->scalar_out_dev;
    const bool has_dist = dist >= 0.0 || dist != dist;      // a negative value = "Euclidean norm" (a NaN length is a given length)
    const int32_t rc = nbk_edge_validity_batch(m, d, d + nq, has_dist ? d + 3 * nq : nullptr, 1, resolution, max_distance, mode, threshold,
                                               reinterpret_cast<uint8_t*>(dout), d + 2 * nq, reinterpret_cast<int32_t*>(dout + 1), mm->scalar_stream);
    if (rc != NBK_OK) return rc;
    NBK_HIP(hipStreamSynchronize(mm->scalar_stream));
    *valid = (int32_t)(reinterpret_cast<const uint8_t*>(mm->scalar_out)[0]);
    if (end != nullptr) memcpy(end, h + 2 * nq, sizeof(double) * nq);
    if (n_samples != nullptr) *n_samples = reinterpret_cast<const int32_t*>(mm->scalar_out + 1)[0];
    return NBK_OK;
}

int32_t nbk_selftest_math(const double* a, const double* b, int64_t n, double* sin_out, double* cos_out, double* sqrt_out,
                          double* div_out, void* stream) {
    if (n < 0 || (n > 0 && (!a || !b || !sin_out || !cos_out || !sqrt_out || !div_out))) return NBK_ERR_INVALID;
    if (n == 0) return NBK_OK;
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, sin_out, cos_out,
                       sqrt_out, div_out);
    NBK_HIP(hipGetLastError());
    return NBK_OK;
}

int32_t nbk_fk_batch_host(const nbk_model* m, const double* q, int64_t B, const int32_t* path, int32_t path_len,
                          const double* local, double* T_out) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || T_out == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (B == 0) return NBK_OK;
    double *dq = nullptr, *dT = nullptr;
    NBK_HIP(hipMalloc((void**)&dq, sizeof(double) * B * m->n_q));
    hipError_t e = hipMalloc((void**)&dT, sizeof(double) * B * 16);
    if (e != hipSuccess) { (void)hipFree(dq); return hip_fail(e, "hipMalloc"); }
    int st = NBK_OK;
    e = hipMemcpy(dq, q, sizeof(double) * B * m->n_q, hipMemcpyHostToDevice);
    if (e != hipSuccess) st = hip_fail(e, "hipMemcpy H2D");
    if (st == NBK_OK) st = nbk_fk_batch(m, dq, B, path, path_len, local, nullptr, dT, nullptr);
    if (st == NBK_OK) { e = hipMemcpy(T_out, dT, sizeof(double) * B * 16, hipMemcpyDeviceToHost); if (e != hipSuccess) st = hip_fail(e, "hipMemcpy D2H"); }
    (void)hipFree(dq); (void)hipFree(dT);
    return st;
}

int32_t nbk_validity_batch_host(const nbk_model* m, const double* q, int64_t B, double threshold, uint8_t* mask_bytes) {
    if (m == nullptr || B < 0 || (B > 0 && (q == nullptr || mask_bytes == nullptr))) return NBK_ERR_INVALID;
    NBK_DEVICE(m);
    if (B == 0) return NBK_OK;
    double* dq = nullptr;
    uint8_t* dm = nullptr;
    NBK_HIP(hipMalloc((void**)&dq, sizeof(double) * B * m->n_q));
    hipError_t e = hipMalloc((void**)&dm, (size_t)B);
    if (e != hipSuccess) { (void)hipFree(dq); return hip_fail(e, "hipMalloc"); }
    int st = NBK_OK;
    e = hipMemcpy(dq, q, sizeof(double) * B * m->n_q, hipMemcpyHostToDevice);
    if (e != hipSuccess) st = hip_fail(e, "hipMemcpy H2D");
    if (st == NBK_OK) st = nbk_validity_batch(m, dq, B, threshold, nullptr, dm, nullptr);
    if (st == NBK_OK) { e = hipMemcpy(mask_bytes, dm, (size_t)B, hipMemcpyDeviceToHost); if (e != hipSuccess) st = hip_fail(e, "hipMemcpy D2H"); }
    (void)hipFree(dq); (void)hipFree(dm);
    return st;
}

}  // extern "C"
