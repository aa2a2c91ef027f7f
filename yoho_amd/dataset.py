"""Evaluation dataset description - drop-in for the test-time half of the reference's utils/dataset.py
(EvalDataset / ThrDMatchPartDataset :21-143, get_dataset_name :146-229, get_dataset :232-238).

The training Dataset classes (:242-323) are in yoho_amd.train.trainer.  open3d is not required: keypoints come from
`Keypoints_PC/cloud_bin_{k}Keypoints.npy` when present (what the hot path reads), otherwise from the point
cloud (.ply parsed by a small numpy reader, or .txt) and the `Keypoints/*.txt` index files exactly as the
reference does.
"""
import os
import numpy as np

from .utils import make_non_exists_dir


def read_ply_xyz(fn):
    """Vertex positions of an ascii / binary_little_endian .ply (enough for 3DMatch fragments)."""
    with open(fn, 'rb') as f:
        fmt, nvert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode('ascii', 'ignore').strip()
            if line.startswith('format'):
                fmt = line.split()[1]
            elif line.startswith('element'):
                in_vertex = line.split()[1] == 'vertex'
                if in_vertex:
                    nvert = int(line.split()[2])
            elif line.startswith('property') and in_vertex:
                props.append((line.split()[-1], line.split()[1]))
            elif line == 'end_header':
                break
            elif line == '':
                raise ValueError(f'{fn}: truncated ply header')
        tmap = {'float': 'f4', 'float32': 'f4', 'double': 'f8', 'float64': 'f8', 'uchar': 'u1', 'uint8': 'u1', 'char': 'i1',
                'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'short': 'i2', 'ushort': 'u2'}
        if fmt == 'ascii':
            data = np.loadtxt(f, max_rows=nvert, ndmin=2)
            names = [p[0] for p in props]
            return np.stack([data[:, names.index(a)] for a in 'xyz'], 1).astype(np.float64)
        if fmt != 'binary_little_endian':
            raise NotImplementedError(f'{fn}: ply format {fmt}')
        dt = np.dtype([(n, '<' + tmap[t]) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * nvert), dtype=dt, count=nvert)
        return np.stack([data['x'], data['y'], data['z']], 1).astype(np.float64)


class EvalDataset:
    pass


class ThrDMatchPartDataset(EvalDataset):
    """utils/dataset.py:55-143"""

    def __init__(self, root_dir, stationnum, gt_dir=None):
        self.root = root_dir
        if gt_dir is None:
            self.gt_dir = f'{self.root}/PointCloud/gt.log'
        else:
            self.gt_dir = gt_dir
        self.kps_pc_fn = [f'{self.root}/Keypoints_PC/cloud_bin_{k}Keypoints.npy' for k in range(stationnum)]
        self.kps_fn = [f'{self.root}/Keypoints/cloud_bin_{k}Keypoints.txt' for k in range(stationnum)]
        self.pc_ply_paths = [f'{self.root}/PointCloud/cloud_bin_{k}.ply' for k in range(stationnum)]
        self.pc_txt_paths = [f'{self.root}/PointCloud/cloud_bin_{k}.txt' for k in range(stationnum)]
        self.pair_id2transform = self.parse_gt_fn(self.gt_dir)
        self.pair_ids = [tuple(v.split('-')) for v in self.pair_id2transform.keys()]
        self.pc_ids = [str(k) for k in range(stationnum)]
        self.pair_num = self.get_pair_nums()
        self.name = '3dmatch/kitchen'

    @staticmethod
    def parse_gt_fn(fn):
        """:74-89 - note: ids and rows are parsed as float32, rows by whitespace (np.fromstring sep=' ')"""
        with open(fn, 'r') as f:
            lines = f.readlines()
        pair_num = len(lines) // 5
        pair_id2transform = {}
        for k in range(pair_num):
            id0, id1 = np.array(lines[k * 5].split()[0:2], dtype=np.float32)
            id0 = int(id0); id1 = int(id1)
            rows = [np.array(lines[k * 5 + r].split(), dtype=np.float32) for r in (1, 2, 3)]
            pair_id2transform['-'.join((str(id0), str(id1)))] = np.stack(rows, 0)
        return pair_id2transform

    def get_pair_ids(self):
        return self.pair_ids

    def get_pair_nums(self):
        return len(self.pair_ids)

    def get_cloud_ids(self):
        return self.pc_ids

    def get_pc_dir(self, cloud_id):
        return self.pc_ply_paths[int(cloud_id)]

    def get_pc(self, pc_id):
        if os.path.exists(self.pc_ply_paths[int(pc_id)]):
            return read_ply_xyz(self.pc_ply_paths[int(pc_id)])
        return np.loadtxt(self.pc_txt_paths[int(pc_id)], delimiter=',')

    def get_key_dir(self, cloud_id):
        return self.kps_fn[int(cloud_id)]

    def get_transform(self, id0, id1):
        return self.pair_id2transform['-'.join((id0, id1))]

    def get_name(self):
        return self.name

    def get_kps(self, cloud_id):
        """:123-143.  Superset: if neither the cloud nor the index file exists but Keypoints_PC/*.npy does
        (the only file the hot path itself reads), that array is returned."""
        k = int(cloud_id)
        have_pc = os.path.exists(self.pc_ply_paths[k]) or os.path.exists(self.pc_txt_paths[k])
        if not have_pc and os.path.exists(self.kps_pc_fn[k]):
            return np.load(self.kps_pc_fn[k])
        if os.path.exists(self.kps_fn[k]):
            pc = self.get_pc(cloud_id)
            key_idxs = np.loadtxt(self.kps_fn[k]).astype(int)
            keys = pc[key_idxs]
            make_non_exists_dir(f'{self.root}/Keypoints_PC')
            np.save(self.kps_pc_fn[k], keys)
            return keys
        pc = self.get_pc(cloud_id)          # random sample 5000
        key_idxs = np.arange(pc.shape[0])
        np.random.shuffle(key_idxs)
        key_idxs = key_idxs[0:5000]
        keys = pc[key_idxs]
        make_non_exists_dir(f'{self.root}/Keypoints')
        np.savetxt(self.kps_fn[k], key_idxs)
        make_non_exists_dir(f'{self.root}/Keypoints_PC')
        np.save(self.kps_pc_fn[k], keys)
        return keys


_3DMATCH = ["kitchen", "sun3d-home_at-home_at_scan1_2013_jan_1", "sun3d-home_md-home_md_scan9_2012_sep_30", "sun3d-hotel_uc-scan3",
            "sun3d-hotel_umd-maryland_hotel1", "sun3d-hotel_umd-maryland_hotel3", "sun3d-mit_76_studyroom-76-1studyroom2",
            "sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika"]
_SCENES = {    # utils/dataset.py:149-208 (scene list, fragments per scene)
    'demo': (['kitchen'], [2]),
    '3dmatch': (_3DMATCH, [60, 60, 60, 55, 57, 37, 66, 38]),
    '3dLomatch': (_3DMATCH, [60, 60, 60, 55, 57, 37, 66, 38]),
    'ETH': (['gazebo_summer', 'gazebo_winter', 'wood_autumn', 'wood_summer'], [32, 31, 32, 37]),
    'WHU-TLS': (['Park', 'Mountain', 'Campus', 'RiverBank', 'UndergroundExcavation', 'Tunnel'], [32, 6, 10, 7, 12, 7]),
}


def get_dataset_name(dataset_name, origin_data_dir):
    """utils/dataset.py:146-229 (test sets; '3dmatch_train' is a training set and out of scope)."""
    if dataset_name not in _SCENES:
        raise NotImplementedError
    scenes, stationnums = _SCENES[dataset_name]
    datasets = {'wholesetname': f'{dataset_name}'}
    for scene, n in zip(scenes, stationnums):
        if dataset_name == '3dLomatch':        # shares 3dmatch's data, own ground truth (:179-182)
            root_dir = f'{origin_data_dir}/3dmatch/' + scene
            ds = ThrDMatchPartDataset(root_dir, n, f'{root_dir}/PointCloud/gtLo.log')
        else:
            root_dir = f'{origin_data_dir}/{dataset_name}/' + scene
            ds = ThrDMatchPartDataset(root_dir, n)
        ds.name = f'{dataset_name}/{scene}'
        datasets[scene] = ds
    return datasets


def get_dataset(cfg, training=True):
    """utils/dataset.py:232-238"""
    dataset_name = cfg.trainset_name if training else cfg.testset_name
    return get_dataset_name(dataset_name, cfg.origin_data_dir)
