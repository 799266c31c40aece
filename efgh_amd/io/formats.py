"""On-disk formats the reference's loaders / drivers use, host side only.

* KITTI-style velodyne `.bin`: float32 x 4 per point                (loader_utils.py:59-61)
* KITTI odometry `poses.txt` lines (12 floats) and `calib.txt`        (loader_utils.py:12-52)
* RELLIS `camera_info.txt` (fx fy cx cy) and lidar->camera yaml       (loader_utils.py:206-229)
* random-initialisation CSV  name,roll,pitch,yaw,tx,ty,tz,cam_roll   (rellis3d_loader.py:44-48)
* prediction CSV  fname,<12 floats of sensor2_T_sensor1[:3,:]>,      (test.py:46-53)
"""
import csv

import numpy as np


def read_velodyne_bin(path):
    """-> (N,4) float32 (x,y,z,reflectance)"""
    return np.fromfile(path, dtype=np.float32).reshape((-1, 4))


def write_velodyne_bin(path, pts):
    np.asarray(pts, dtype=np.float32).reshape((-1, 4)).tofile(path)


def parse_pose_line(line):
    """one line of KITTI poses.txt -> 4x4"""
    v = np.array([float(p) for p in line.split()], dtype=float).reshape((3, 4))
    out = np.eye(4)
    out[:3, :] = v
    return out


def read_kitti_calib(path):
    data = {}
    with open(path) as f:
        for line in f:
            if ':' not in line:
                continue
            key, value = line.split(':', 1)
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                pass
    P2, Tr = np.eye(4), np.eye(4)
    P2[:3, :] = data['P2'].reshape(3, 4)
    Tr[:3, :] = data['Tr'].reshape(3, 4)
    return {'Tr': Tr, 'Tr_inv': np.linalg.inv(Tr), 'P2': P2, 'P2_inv': np.linalg.inv(P2)}


def read_rellis_camera_info(path):
    d = np.loadtxt(path)
    K = np.zeros((3, 3))
    K[0, 0], K[1, 1], K[2, 2], K[0, 2], K[1, 2] = d[0], d[1], 1, d[2], d[3]
    return K


def quat_xyzw_to_matrix(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def read_rellis_lidar2cam(path):
    import yaml
    with open(path) as f:
        d = yaml.safe_load(f)['os1_cloud_node-pylon_camera_node']
    RT = np.eye(4)
    RT[:3, :3] = quat_xyzw_to_matrix(np.array([d['q']['x'], d['q']['y'], d['q']['z'], d['q']['w']], dtype=float))
    RT[:3, 3] = [d['t']['x'], d['t']['y'], d['t']['z']]
    return np.linalg.inv(RT)


def read_rand_init_csv(path):
    """-> {name: [roll, pitch, yaw, tx, ty, tz, cam_roll]} (radians / metres)"""
    out = {}
    with open(path) as f:
        for line in csv.reader(f):
            if line:
                out[line[0]] = [float(v) for v in line[1:]]
    return out


def append_prediction_csv(path, fname, sensor2_T_sensor1):
    """one row per sample, exactly as test.py:46-53 writes it (trailing comma included)"""
    v = np.asarray(sensor2_T_sensor1, dtype=np.float32)[:3, :].flatten()
    with open(path, 'a') as f:
        f.write(fname + ',' + ''.join(str(x) + ',' for x in v) + '\n')


def read_prediction_csv(path):
    out = {}
    with open(path) as f:
        for line in csv.reader(f):
            if line:
                out[line[0]] = np.array([float(v) for v in line[1:13]], dtype=np.float32).reshape(3, 4)
    return out
