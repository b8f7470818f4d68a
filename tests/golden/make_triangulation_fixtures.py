"""Writes tests/golden/triangulation_reference_fixtures.npz from the reference's OWN triangulation test data
(test/triangulation.cpp): the pose vectors, feature tracks, camera matrices and Matlab values of the test cases
"visual" (:56-246), "stereo_visual" (:248-474), "pinv" (:477-485) and "der_triangulateWithTwoCameras" (:521-580).
These are data fixtures (numbers), parsed here so the tests also run where /root/reference is absent (the GPU box).
Run in the build container:  python tests/golden/make_triangulation_fixtures.py
"""
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "triangulation_reference_fixtures.npz")
num = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def numbers(text):
    return np.array([float(x) for x in re.findall(num, text)])


src = open(os.path.join(REF, "test/triangulation.cpp")).read()
visual, stereo = src[src.index('TEST_CASE( "visual"'):src.index('TEST_CASE( "stereo_visual"')], \
    src[src.index('TEST_CASE( "stereo_visual"'):src.index('TEST_CASE( "pinv"')]
v_poses = numbers(re.search(r"poses; poses <<(.*?);", visual, re.S).group(1))
v_uv = numbers(re.search(r"uv; uv <<(.*?);", visual, re.S).group(1)).reshape(10, 2)
v_pf = numbers(re.search(r"pf_e\((.*?)\)", visual).group(1))
s_poses = numbers(re.search(r"poses;\s*poses <<(.*?);", stereo, re.S).group(1))
s_uv = numbers(re.search(r" uv; uv <<(.*?);", stereo, re.S).group(1)).reshape(10, 2)
s_uv2 = numbers(re.search(r"secondUv; secondUv <<(.*?);", stereo, re.S).group(1)).reshape(10, 2)
s_imu = numbers(re.search(r"params\.odometry\.imuToCameraMatrix = \{(.*?)\}", stereo, re.S).group(1))
s_imu2 = numbers(re.search(r"params\.odometry\.secondImuToCameraMatrix = \{(.*?)\}", stereo, re.S).group(1))
assert v_poses.shape == (70,) and s_poses.shape == (70,) and v_pf.shape == (3,) and s_imu.shape == (9,) and s_imu2.shape == (9,)
pinv_case = src[src.index('TEST_CASE( "pinv"'):src.index('TEST_CASE( "triangulateWithTwoCameras"')]
pinv_m = numbers(re.search(r"m; m <<(.*?);", pinv_case).group(1)).reshape(2, 3)
pinv_e = numbers(re.search(r"pinvm <<(.*?);", pinv_case, re.S).group(1)).reshape(3, 2)
two = src[src.index('TEST_CASE( "der_triangulateWithTwoCameras"'):src.index('TEST_CASE( "der_inverseDepth"')]
ip0 = numbers(re.search(r"ip0\((.*?)\)", two).group(1)); ip1 = numbers(re.search(r"ip1\((.*?)\)", two).group(1))
two_p0, two_q0, two_p1, two_q1 = (numbers(re.search(p, two).group(1)) for p in (
    r"segment\(0, 3\) = Eigen::Vector3d\((.*?)\)", r"segment\(3, 4\) = Eigen::Vector4d\((.*?)\)",
    r"segment\(7, 3\) = Eigen::Vector3d\((.*?)\)", r"segment\(10, 4\) =\s*Eigen::Vector4d\((.*?)\)"))
inv_p0 = numbers(re.search(r"p0\((.*?)\);", src[src.index('TEST_CASE( "der_inverseDepth"'):]).group(1))
# codegen/parameter_definitions.c:178,187: default imuToCameraMatrix and stereoCameraTranslation
defs = open(os.path.join(REF, "codegen/parameter_definitions.c")).read()
imu_default = numbers(re.search(r"odometry\.imuToCameraMatrix (.*)", defs).group(1))
stereo_translation = numbers(re.search(r"odometry\.stereoCameraTranslation (.*)", defs).group(1))
np.savez_compressed(OUT, visual_poses=v_poses, visual_uv=v_uv, visual_pf_matlab=v_pf, stereo_poses=s_poses, stereo_uv=s_uv,
                    stereo_uv2=s_uv2, stereo_imu=s_imu, stereo_imu2=s_imu2, pinv_m=pinv_m, pinv_matlab=pinv_e, two_ip0=ip0, two_ip1=ip1,
                    two_p0=two_p0, two_q0=two_q0, two_p1=two_p1, two_q1=two_q1, inverse_depth_p0=inv_p0, imu_default=imu_default,
                    stereo_translation=stereo_translation)
print("wrote", OUT, v_pf, two_p0, two_q1, inv_p0, imu_default, stereo_translation)
