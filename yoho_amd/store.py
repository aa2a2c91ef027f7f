"""HBM-resident stage store.

The reference passes every stage through .npy files (~300 MB of reads per
pair).  The drop-in classes keep writing those files byte-compatibly, but they also park the
device tensors here so that the next stage of the same process reads HBM instead of disk.
With 288 GB per MI355X a whole 3DMatch scene (<= 66 fragments x 2 x 38.4 MB) stays resident.
"""
import os
import numpy as np
import torch

_store = {}
_bytes = 0
_stamp = {}            # path key -> (mtime_ns, size) of the file the resident tensor mirrors
_cap = None


def _max_bytes():
    """YOHO_STORE_BYTES, else half of the device memory that is free when the store is first used"""
    global _cap
    if _cap is None:
        env = os.environ.get("YOHO_STORE_BYTES")
        if env:
            _cap = int(env)
        elif torch.cuda.is_available():
            _cap = int(torch.cuda.mem_get_info()[0] // 2)
        else:
            _cap = 8 << 30
    return _cap


def _file_stamp(path):
    try:
        st = os.stat(path)
        return (st.st_mtime_ns, st.st_size)
    except OSError:
        return None


def put(key, tensor):
    global _bytes
    if key in _store:
        _bytes -= _store[key].numel() * _store[key].element_size()
    nb = tensor.numel() * tensor.element_size()
    while _store and _bytes + nb > _max_bytes():
        k, v = next(iter(_store.items()))
        _bytes -= v.numel() * v.element_size()
        del _store[k]
        _stamp.pop(k, None)
    _store[key] = tensor
    _bytes += nb


def get(key):
    return _store.get(key)


def clear():
    global _bytes
    _store.clear()
    _stamp.clear()
    _bytes = 0


def load_npy(path, dtype=torch.float32):
    """Device tensor for a cached .npy stage file: the resident copy if it still mirrors the file on disk (same mtime
    and size as when it was stored - another process may have regenerated the stage), else the file."""
    key = os.path.abspath(path)
    t = get(key)
    if t is not None and _stamp.get(key) != _file_stamp(key):
        t = None
    if t is None:
        t = torch.from_numpy(np.ascontiguousarray(np.load(path))).to(device="cuda", dtype=dtype)
        put(key, t)
        _stamp[key] = _file_stamp(key)
    return t


def save_npy(path, tensor):
    """write the stage file (atomically: ranks of a sharded run may produce the same fragment's file) and keep the tensor"""
    key = os.path.abspath(path)
    tmp = f"{key}.{os.getpid()}.tmp.npy"
    np.save(tmp, tensor.cpu().numpy())
    os.replace(tmp, key)
    put(key, tensor)
    _stamp[key] = _file_stamp(key)
