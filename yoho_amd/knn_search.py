"""utils/knn_search.py mirror: brute-force nearest neighbour on the GPU (yoho_nn_search).

``knn_module.KNN(1)(target_F (1,f,n), source_F (1,f,m)) -> (d (1,1,m), idx (1,1,m))`` as the
reference; k > 1 is not on the hot path (the reference only ever builds KNN(1):
tests/matcher.py:17, YOHO_testset.py:66) and raises NotImplementedError."""
import numpy as np
import torch

from . import hip


class modified_knn_matcher():
    def __init__(self, k=1):
        self.k = k
        self.ctx = hip.get_context()

    def _prep(self, F):
        F = F.squeeze()
        if F.dim() == 1:
            F = F[None]
        return F.to(device="cuda", dtype=torch.float32).contiguous()

    def find_nn_gpu(self, source_F, target_F, nn_max_n=1000, return_distance=True, dist_type='SquareL2'):
        """utils/knn_search.py:26-66.  Returns (dists, inds) - in that order, as the reference."""
        if dist_type not in ("L2", "SquareL2"):
            raise NotImplementedError('Not implemented')
        F0, F1 = self._prep(source_F), self._prep(target_F)
        if F0.shape[1] not in (3, 32):
            raise NotImplementedError(f"feature width {F0.shape[1]} (the path uses 32-D descriptors and 3-D points)")
        d, inds = self.ctx.nn_search(F0, F1, want_dist=True, squared=(dist_type == "SquareL2"))
        dists, inds = d.cpu(), inds.cpu()
        return (dists, inds) if return_distance else inds

    def find_corr(self, F0, F1, subsample_size=-1, mutual=True, nn_max_n=500):
        """utils/knn_search.py:106-136"""
        inds0, inds1 = np.arange(F0.shape[0]), np.arange(F1.shape[0])
        if subsample_size > 0:
            N0, N1 = min(len(F0), subsample_size), min(len(F1), subsample_size)
            inds0 = np.random.choice(len(F0), N0, replace=False)
            inds1 = np.random.choice(len(F1), N1, replace=False)
            F0, F1 = F0[inds0], F1[inds1]
        if not mutual:
            nn = self.find_nn_gpu(F0, F1, return_distance=False).numpy()
            return inds0, inds1[nn]
        m = self.ctx.mutual_nn(self._prep(F0), self._prep(F1)).cpu().numpy()
        return inds0[m[:, 0]], inds1[m[:, 1]]

    def __call__(self, target_F, source_F, nn_max_n=500, dist_type='L2'):
        """utils/knn_search.py:138-161: target_F 1*f*n, source_F 1*f*m -> d, idx of shape 1*1*m."""
        if self.k >= 2:
            raise NotImplementedError("k > 1 is not used on the YOHO hot path")
        tgt = target_F.squeeze().T
        src = source_F.squeeze().T
        d, idx = self.find_nn_gpu(source_F=src, target_F=tgt, nn_max_n=nn_max_n, return_distance=True, dist_type=dist_type)
        return d[None, None], idx[None, None]


class knn_module_class():
    def KNN(self, k):
        return modified_knn_matcher(k)


knn_module = knn_module_class()
