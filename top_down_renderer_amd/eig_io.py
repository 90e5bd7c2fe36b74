"""The reference's on-disk map cache format (include/top_down_render/top_down_map.h:29-50, files
~/.ros/xview_cache/{class_map<i>,geo_map<i>,class_mask}.eig written by src/top_down_map.cpp:263-286):
`Index rows, Index cols` (2 x int64) followed by the column-major raw scalars of an Eigen array."""
import os

import numpy as np


def write_eig(path, array):
    """array indexed [row, col] (float32 distance map or uint8 mask)."""
    a = np.asarray(array)
    if a.ndim != 2:
        raise ValueError("an Eigen array is 2-D")
    with open(path, "wb") as f:
        np.asarray([a.shape[0], a.shape[1]], "<i8").tofile(f)
        np.asfortranarray(a).ravel(order="F").tofile(f)


def read_eig(path, dtype):
    """Returns the array indexed [row, col]."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        hdr = np.fromfile(f, "<i8", 2)
        if len(hdr) != 2 or hdr[0] < 0 or hdr[1] < 0:
            raise ValueError(f"{path}: bad .eig header")
        rows, cols = int(hdr[0]), int(hdr[1])
        if 16 + rows * cols * np.dtype(dtype).itemsize != size:
            raise ValueError(f"{path}: size does not match a {rows}x{cols} array of {np.dtype(dtype)}")
        data = np.fromfile(f, dtype, rows * cols)
    return data.reshape(cols, rows).T.copy()


def load_cached_maps(cache_dir, num_classes):
    """loadCachedMaps (src/top_down_map.cpp:244-261): (class_maps (ncls, H, W) f32, class_mask (H, W) u8)."""
    maps = np.stack([read_eig(os.path.join(cache_dir, f"class_map{c}.eig"), np.float32) for c in range(num_classes)])
    mask = read_eig(os.path.join(cache_dir, "class_mask.eig"), np.uint8)
    return maps, mask


def save_cached_maps(cache_dir, map_path, class_maps, class_mask, resolution):
    """saveCachedMaps (src/top_down_map.cpp:263-286) minus the geo maps (dead on the scoring path)."""
    os.makedirs(cache_dir, exist_ok=True)
    with open(os.path.join(cache_dir, "cached_data.txt"), "w") as f:
        f.write(f"{map_path}\n{len(class_maps)}\n{resolution:g}\n")
    for c, m in enumerate(class_maps):
        write_eig(os.path.join(cache_dir, f"class_map{c}.eig"), np.asarray(m, np.float32))
    write_eig(os.path.join(cache_dir, "class_mask.eig"), np.asarray(class_mask, np.uint8))
