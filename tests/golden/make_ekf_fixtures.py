"""Writes tests/golden/ekf_reference_fixtures.npz from the reference's OWN EKF test data:
  test/data/P.csv (55x55), test/data/m.csv (55)              -- transformTo round trip (test/ekf.cpp:119-145)
  the 20x20 matrix M, vector v and Matlab value 1.7626        -- chi-squared LDLT test (test/ekf.cpp:19-71)
  the 70-vector of poses, gyro/acc sample, t = dt = 0.01      -- der_predict (test/ekf.cpp:73-117)
  q, rmat_e, dR_e Matlab goldens                              -- test/util.cpp:9-60
These are data fixtures (numbers), parsed here so the tests also run where /root/reference is
absent (the GPU box).  Run in the build container:  python tests/golden/make_ekf_fixtures.py
"""
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ekf_reference_fixtures.npz")
num = r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?"


def numbers(text):
    return np.array([float(x) for x in re.findall(num, text)])


ekf_cpp = open(os.path.join(REF, "test/ekf.cpp")).read()
blk = re.findall(r"M\.block\(0, \d+, 20, 10\) <<(.*?);", ekf_cpp, re.S)
M = np.hstack([numbers(b).reshape(20, 10) for b in blk]) * 1e3
v = numbers(re.search(r"v <<(.*?);", ekf_cpp, re.S).group(1))
poses = numbers(re.search(r"poses; poses <<(.*?);", ekf_cpp, re.S).group(1))
gyro = numbers(re.search(r"gyro; gyro <<(.*?);", ekf_cpp).group(1))
acc = numbers(re.search(r"acc; acc <<(.*?);", ekf_cpp).group(1))
assert M.shape == (20, 20) and v.shape == (20,) and poses.shape == (70,)

util_cpp = open(os.path.join(REF, "test/util.cpp")).read()
q = numbers(re.search(r"Eigen::Vector4d q; q <<(.*?);", util_cpp).group(1))
rmat_e = numbers(re.search(r"rmat_e <<(.*?);", util_cpp, re.S).group(1)).reshape(3, 3)
dR_e = np.stack([numbers(b).reshape(3, 3) for b in re.findall(r"dR_e\[\d\] <<(.*?);", util_cpp, re.S)])
assert dR_e.shape == (4, 3, 3)

P = np.loadtxt(os.path.join(REF, "test/data/P.csv"), delimiter=",")
m = np.loadtxt(os.path.join(REF, "test/data/m.csv"), delimiter=",")
np.savez_compressed(OUT, chi2_M=M, chi2_v=v, chi2_matlab=1.7626, poses=poses, gyro=gyro, acc=acc,
                    q=q, rmat_e=rmat_e, dR_e=dR_e, P55=P, m55=m)
print(OUT, os.path.getsize(OUT), "bytes", M.shape, P.shape, m.shape)
