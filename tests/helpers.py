"""Test helpers: hand-built KinematicModels for the synthetic golden chains."""
import numpy as np

from numbotics_amd.robots.model import KinematicModel, FrameRef, _T34, _skew

REV, PRI, SPH, FIX = 0, 1, 2, 4


def kin_from_sequence(offsets, axes, types, idxs, base, n_q, trailing=None):
    """Merge FIXED entries into the next moving joint exactly like arm.py:38-52, then pack."""
    acc = np.eye(4)
    merged, ax, ty, qi = [], [], [], []
    for off, a, t, i in zip(offsets, axes, types, idxs):
        acc = acc @ off
        if t == FIX:
            continue
        merged.append(acc)
        ax.append(a)
        ty.append(0 if t == REV else 1)
        qi.append(int(i))
        acc = np.eye(4)
    local = acc if trailing is None else acc @ trailing
    J = len(merged)
    rot = np.zeros((J, 27)); trans = np.zeros((J, 3)); slide = np.zeros((J, 3))
    for k in range(J):
        R = merged[k][:3, :3]
        trans[k] = merged[k][:3, 3]
        if ty[k] == 0:
            K = np.outer(ax[k], ax[k])
            rot[k, 0:9] = (R @ K).ravel(); rot[k, 9:18] = (R @ (K - np.eye(3))).ravel()
            rot[k, 18:27] = (R @ _skew(ax[k])).ravel()
        else:
            rot[k, 0:9] = R.ravel(); slide[k] = R @ ax[k]
    rot, trans, slide = rot + 0.0, trans + 0.0, slide + 0.0          # exact zeros as +0.0, like compile_kinematics
    km = KinematicModel(
        n_q=n_q, joint_parent=np.arange(-1, J - 1, dtype=np.int32), joint_type=np.array(ty, dtype=np.int32),
        joint_qidx=np.array(qi, dtype=np.int32), joint_offset=np.array([_T34(m) for m in merged]).reshape(J, 12),
        joint_axis=np.array(ax).reshape(J, 3), joint_rot=rot, joint_trans=trans, joint_slide=slide,
        base_pose=_T34(base), link_names=["end"])
    km.frames["end"] = FrameRef(joint=J - 1, local=local, path=np.arange(J, dtype=np.int32), trailing_fixed=True)
    return km
