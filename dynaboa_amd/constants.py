"""Numeric constants consumed by the adaptation hot path.

Mirrors the values the reference reads from ``constants.py:1-2,6-7,15-98`` and
``config.py:14-17`` (focal length, crop size, the 49-joint naming scheme and the
joint selectors).  The tables are rebuilt here from the joint *names* so that the
49-entry gather used by the SMPL wrapper (reference ``model/smpl.py:20,30``) is
derived, not pasted.
"""

FOCAL_LENGTH = 5000.0
IMG_RES = 224
IMG_NORM_MEAN = (0.485, 0.456, 0.406)
IMG_NORM_STD = (0.229, 0.224, 0.225)

NUM_VERTS = 6890
NUM_SMPL_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_FEATS = 207          # 23 body joints * 9
NUM_EXTRA_JOINTS = 9          # rows of J_regressor_extra
NUM_OUT_JOINTS = 49           # 25 OpenPose-style + 24 ground-truth-style

# SMPL kinematic tree.
SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14,
                16, 17, 18, 19, 20, 21)

# smplx VertexJointSelector vertex ids, in the order it appends them
# (face, feet, finger tips) -> joints 24..44 of the 45-joint smplx output.
VERTEX_JOINT_IDS = (
    332, 6260, 2800, 4071, 583,                     # nose, reye, leye, rear, lear
    3216, 3226, 3387, 6617, 6624, 6787,             # L big/small toe, L heel, R big/small toe, R heel
    2746, 2319, 2445, 2556, 2673,                   # left finger tips
    6191, 5782, 5905, 6016, 6133,                   # right finger tips
)

_OPENPOSE = ['Nose', 'Neck', 'RShoulder', 'RElbow', 'RWrist', 'LShoulder', 'LElbow',
             'LWrist', 'MidHip', 'RHip', 'RKnee', 'RAnkle', 'LHip', 'LKnee', 'LAnkle',
             'REye', 'LEye', 'REar', 'LEar', 'LBigToe', 'LSmallToe', 'LHeel',
             'RBigToe', 'RSmallToe', 'RHeel']
_GT24 = ['Right Ankle', 'Right Knee', 'Right Hip', 'Left Hip', 'Left Knee', 'Left Ankle',
         'Right Wrist', 'Right Elbow', 'Right Shoulder', 'Left Shoulder', 'Left Elbow',
         'Left Wrist', 'Neck (LSP)', 'Top of Head (LSP)', 'Pelvis (MPII)', 'Thorax (MPII)',
         'Spine (H36M)', 'Jaw (H36M)', 'Head (H36M)', 'Nose', 'Left Eye', 'Right Eye',
         'Left Ear', 'Right Ear']
JOINT_NAMES = ['OP ' + n for n in _OPENPOSE] + _GT24

# name -> index into the 54-joint list [24 SMPL | 21 vertex joints | 9 regressed extras]
_SMPL_IDX = dict(pelvis=0, lhip=1, rhip=2, lknee=4, rknee=5, lankle=7, rankle=8, neck=12,
                 lshoulder=16, rshoulder=17, lelbow=18, relbow=19, lwrist=20, rwrist=21)
_VJ = {n: 24 + i for i, n in enumerate(
    ['nose', 'reye', 'leye', 'rear', 'lear', 'lbigtoe', 'lsmalltoe', 'lheel',
     'rbigtoe', 'rsmalltoe', 'rheel'])}
_EXTRA = {n: 45 + i for i, n in enumerate(
    ['Right Hip', 'Left Hip', 'Neck (LSP)', 'Top of Head (LSP)', 'Pelvis (MPII)',
     'Thorax (MPII)', 'Spine (H36M)', 'Jaw (H36M)', 'Head (H36M)'])}
JOINT_MAP = {
    'OP Nose': _VJ['nose'], 'OP Neck': _SMPL_IDX['neck'],
    'OP RShoulder': _SMPL_IDX['rshoulder'], 'OP RElbow': _SMPL_IDX['relbow'],
    'OP RWrist': _SMPL_IDX['rwrist'], 'OP LShoulder': _SMPL_IDX['lshoulder'],
    'OP LElbow': _SMPL_IDX['lelbow'], 'OP LWrist': _SMPL_IDX['lwrist'],
    'OP MidHip': _SMPL_IDX['pelvis'], 'OP RHip': _SMPL_IDX['rhip'],
    'OP RKnee': _SMPL_IDX['rknee'], 'OP RAnkle': _SMPL_IDX['rankle'],
    'OP LHip': _SMPL_IDX['lhip'], 'OP LKnee': _SMPL_IDX['lknee'],
    'OP LAnkle': _SMPL_IDX['lankle'], 'OP REye': _VJ['reye'], 'OP LEye': _VJ['leye'],
    'OP REar': _VJ['rear'], 'OP LEar': _VJ['lear'], 'OP LBigToe': _VJ['lbigtoe'],
    'OP LSmallToe': _VJ['lsmalltoe'], 'OP LHeel': _VJ['lheel'],
    'OP RBigToe': _VJ['rbigtoe'], 'OP RSmallToe': _VJ['rsmalltoe'],
    'OP RHeel': _VJ['rheel'],
    'Right Ankle': _SMPL_IDX['rankle'], 'Right Knee': _SMPL_IDX['rknee'],
    'Left Knee': _SMPL_IDX['lknee'], 'Left Ankle': _SMPL_IDX['lankle'],
    'Right Wrist': _SMPL_IDX['rwrist'], 'Right Elbow': _SMPL_IDX['relbow'],
    'Right Shoulder': _SMPL_IDX['rshoulder'], 'Left Shoulder': _SMPL_IDX['lshoulder'],
    'Left Elbow': _SMPL_IDX['lelbow'], 'Left Wrist': _SMPL_IDX['lwrist'],
    'Nose': _VJ['nose'], 'Left Eye': _VJ['leye'], 'Right Eye': _VJ['reye'],
    'Left Ear': _VJ['lear'], 'Right Ear': _VJ['rear'],
}
JOINT_MAP.update(_EXTRA)
# 49 indices into the 54-joint list, in JOINT_NAMES order.
JOINT_MAP_49 = tuple(JOINT_MAP[n] for n in JOINT_NAMES)

# 14 LSP joints out of the 17 H36M-regressed joints / out of the 24 GT joints.
H36M_TO_J17 = (6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9)
H36M_TO_J14 = H36M_TO_J17[:14]
J24_TO_J17 = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 18, 14, 16, 17)
J24_TO_J14 = J24_TO_J17[:14]
