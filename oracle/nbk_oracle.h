/*
 * oracle/nbk_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C float64 restatement of the reference's hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; numbotics_amd/ never does.
 *
 * What it restates (reference = landonclark97/numbotics, paths relative to /root/reference):
 *   - FK chain sweep            numbotics/robots/helpers.py:33-113, numbotics/robots/arm.py:369-410
 *   - geometric Jacobian        numbotics/robots/helpers.py:117-187, numbotics/robots/arm.py:413-461
 *   - validity predicate        numbotics/robots/arm.py:599-604  (min signed distance < threshold, strict)
 *   - edge discretisation       numbotics/planning/sampling_based/connectors.py:57-100,
 *                               numbotics/planning/trajectories.py:6-22
 *   - MESH shapes               numbotics/utils/mesh.py:18-37, numbotics/utils/shape.py:81-94: one convex hull per mesh
 *                               object (what GEOM_MESH without the concave flag is in Bullet); hull building itself is host
 *                               Python (numbotics_amd/utils/mesh.py), the oracle consumes vertex / face-plane tables
 *   - link-pair signed distance pybullet 3.2.7 getClosestPoints (numbotics/physics/chain.py:944-951);
 *                               Bullet's source is NOT under /root/reference and pybullet is not
 *                               installed: PARITY UNPINNED for every distance value.  The semantics
 *                               are this build's (DESIGN.md "distance semantics").
 * Pinning: FK / Jacobian / edge sampling are checked against the .npz files under tests/golden/, which were produced
 * by executing the reference's own source (tests/golden/make_golden.py).
 *
 * The arithmetic follows ONE written operation order (DESIGN.md "arithmetic contract": explicit
 * fma(), no implicit contraction, own sincos) so that the HIP kernels can be compared bit-for-bit.
 */
#ifndef NBK_ORACLE_H
#define NBK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SPHERE = 0, ORC_CAPSULE = 1, ORC_BOX = 2, ORC_CYLINDER = 3, ORC_PLANE = 4, ORC_HULL = 5 };
enum { ORC_REVOLUTE = 0, ORC_PRISMATIC = 1 };

typedef struct {
    int32_t n_q;
    int32_t n_joints;
    const int32_t *joint_parent;  /* [J]   -1 = base frame */
    const int32_t *joint_type;    /* [J]   ORC_REVOLUTE / ORC_PRISMATIC */
    const int32_t *joint_qidx;    /* [J] */
    const double *joint_rot;      /* [J][27] M0, M1, M2 (row-major 3x3 each) */
    const double *joint_trans;    /* [J][3] */
    const double *joint_slide;    /* [J][3] */
    const double *joint_axis;     /* [J][3] */
    const double *base_pose;      /* [12] 3x4 row-major */
    int32_t n_rshapes;
    const int32_t *rshape_frame;  /* [S] moving frame, -1 = base */
    const int32_t *rshape_type;   /* [S] */
    const double *rshape_local;   /* [S][12] */
    const double *rshape_param;   /* [S][4] */
    int32_t n_wshapes;
    const int32_t *wshape_type;   /* [W] */
    const double *wshape_pose;    /* [W][12] world */
    const double *wshape_param;   /* [W][4] */
    int32_t n_pairs;
    const int32_t *pair_a;        /* [P] robot shape */
    const int32_t *pair_b;        /* [P] robot shape, or S + world shape */
    /* convex hulls of MESH shapes (numbotics/utils/shape.py:81-94, numbotics/utils/mesh.py:18-37): a shape of type ORC_HULL
     * names its hull in param[0]; vertices / face planes are in the primitive's local frame, whose origin is the mean of
     * the hull's vertices; param[3] = margin (inflates the hull, as Bullet's margin does for btConvexHullShape) */
    int32_t n_hulls;
    const int32_t *hull_vert_begin;  /* [H+1] */
    const double *hull_verts;        /* [NV][3] */
    const int32_t *hull_face_begin;  /* [H+1] */
    const double *hull_planes;       /* [NF][4] unit outward normal n and offset d: inside n.x <= d */
} orc_model;

void orc_sincos(double x, double *s, double *c);
void orc_sincos_array(const double *x, int64_t n, double *s, double *c);
/* elementwise sqrt and divide, for checking that the device rounds them like the host */
void orc_sqrt_div_array(const double *a, const double *b, int64_t n, double *sq, double *dv);

/* FK of one frame: path = joint indices root->frame; local = constant 3x4 after the last joint;
 * local_pose (optional, [B][16]) is right-multiplied per configuration.  out: [B][16] row-major 4x4. */
int orc_fk(const orc_model *m, const double *q, int64_t B, const int32_t *path, int32_t path_len,
           const double *local, const double *local_pose, double *out);

/* mode 0: end pose = T*local; 1: T*local*pose[b]; 2: end position taken from pose[b] (global_pose).
 * out: [B][6][n_q]. */
int orc_jacobian(const orc_model *m, const double *q, int64_t B, const int32_t *path, int32_t path_len,
                 const double *local, int32_t mode, const double *pose, double *out);

/* Arm.inverse_kinematics (numbotics/robots/arm.py:464-552) for B (pose, q0) problems of one frame; limits (optional)
 * [n_q][2]; success[b] = |diff| < tol.  The damped 6x6 system is solved by an unpivoted Cholesky (the reference calls
 * np.linalg.solve): equal to rounding, not bit for bit -- tests/test_oracle_ik.py holds the NumPy restatement. */
int orc_ik(const orc_model *m, const double *pose, const double *q0, int64_t B, const int32_t *path, int32_t path_len,
           const double *local, const double *limits, double tol, int32_t max_iter, int32_t max_failures,
           double *q_out, uint8_t *success, double *diff_norm, int32_t *iters);

/* signed distance of every allowed pair: dist [B][P]; witness (optional) [B][P][9] =
 * point on A, point on B, unit normal from B to A. */
int orc_pair_distances(const orc_model *m, const double *q, int64_t B, double *dist, double *witness);
/* Arm.jacobian_proximity (numbotics/robots/arm.py:620-632) for every allowed pair: dist, witness as above and
 * jrows [B][P][n_q] = n . Jv_subject(point on A) - n . Jv_target(point on B) (world targets: no second term), Jv =
 * linear rows of orc_jacobian mode 2 (numbotics/robots/helpers.py:117-187). */
int orc_proximity_jacobian(const orc_model *m, const double *q, int64_t B, double *dist, double *witness, double *jrows);
/* min over pairs and its index (first minimum).  P == 0 -> +inf / -1. */
int orc_closest(const orc_model *m, const double *q, int64_t B, double *min_dist, int32_t *argmin);
/* mask[b] = (min_p dist < threshold) ? 1 : 0 ; nthreads <= 1 runs serially */
int orc_validity(const orc_model *m, const double *q, int64_t B, double threshold, uint8_t *mask, int32_t nthreads);

/* DiscreteConnector over E edges.  mode 0 = connect, 1 = steer.  dist (optional [E]) overrides the
 * Euclidean norm.  valid[e] = 1 iff every sample is collision free (NOT in_collision(q, threshold));
 * 0 also when dist <= float32 eps (the reference returns None).  end [E][n_q] = goal (connect) or
 * traj(T_f) (steer); n_samples [E] = len(T) (0 for the degenerate case). */
int orc_edge_validity(const orc_model *m, const double *starts, const double *goals, const double *dist,
                      int64_t E, double resolution, double max_distance, int32_t mode, double threshold,
                      uint8_t *valid, double *end, int32_t *n_samples, int32_t nthreads);
/* the sample points of one edge (for checking against golden G5): out [max_samples][n_q]; returns count */
int orc_edge_samples(int32_t n_q, const double *start, const double *goal, double dist, double resolution,
                     double max_distance, int32_t mode, double *out, int32_t max_samples);

/* one shape pair in world coordinates (unit tests of the narrowphase).  pose: 3x4, param: [4].
 * returns signed distance; witness[9] as above; iters (optional) = GJK iterations used (0 if closed form). */
double orc_shape_distance(int32_t type_a, const double *pose_a, const double *param_a,
                          int32_t type_b, const double *pose_b, const double *param_b,
                          double *witness, int32_t *iters);

/* the same with hull shapes: `hulls` carries the hull tables (only the hull_* fields are read) */
double orc_shape_distance_m(const orc_model *hulls, int32_t type_a, const double *pose_a, const double *param_a,
                            int32_t type_b, const double *pose_b, const double *param_b, double *witness, int32_t *iters);
int orc_shape_collides_m(const orc_model *hulls, int32_t type_a, const double *pose_a, const double *param_a, int32_t type_b,
                         const double *pose_b, const double *param_b, double threshold);

/* statistics of the predicate since the last reset, out[5]: items, survivors of the sphere test, GJK calls, inflated walks,
 * inflated walks left undecided (per thread: meaningful after single-threaded runs only) */
void orc_stats(long long *out, int reset);
/* out[120]: iterations of the distance predicate [0..79], of the inflated walk [80..118], undecided inflated walks [119] */
void orc_pred_hist(long long *out, int reset);
/* diagnostic: the cores of pair `p` at configuration q, the distance both ways round and the walk of both predicates, to stderr
 * (tools/fuzz_repro.py, Oracle.pair_trace) */
int orc_pair_trace(const orc_model *m, const double *q, int32_t p, double thr);

/* the validity predicate of one pair: signed distance < threshold, decided with early outs */
int orc_shape_collides(int32_t type_a, const double *pose_a, const double *param_a, int32_t type_b,
                       const double *pose_b, const double *param_b, double threshold);

#ifdef __cplusplus
}
#endif
#endif
